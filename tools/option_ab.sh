#!/bin/bash
# A/B of one library option on ONE box under tools/ab.py's protocol (>= 5 alternations, median and min - max per arm, no winner
# when the intervals overlap).   bash tools/option_ab.sh pair_waves 1536 2048 [rounds] [--diag]
cd "$(dirname "$0")/.."
opt=${1:?option}; a=${2:?value A}; b=${3:?value B}; rounds=${4:-5}; shift 4 2>/dev/null || shift $#
exec python3 tools/ab.py --option "$opt" "$a" "$b" --rounds "$rounds" "$@"

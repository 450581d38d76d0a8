"""sift_amd — MI355X-native drop-in for snowiow/SIFT's Sift::calculate() hot path."""

"""Independent restatement (test infrastructure; plain Python, no library call) of what /root/reference/main.cpp:59-76 draws:
for every returned InterestPoint a cv::RotatedRect of side (int)(10 * scale) at (loc * 2^octave) / subpixel_divisor (both
stored in u16_t), its four corners (cv::RotatedRect::points, OpenCV 3.2 modules/core/src/matrix.cpp) joined by cv::line with
the defaults (thickness 1, lineType 8, shift 0, colour Scalar(255, 0, 0) on a B,G,R image): corner coordinates rounded by
cvRound (nearest, ties to even), the segment clipped by cv::clipLine when an end lies outside the image, then walked by
cv::LineIterator (connectivity 8, left to right) FROM THE CLIPPED END POINTS (modules/imgproc/src/drawing.cpp).

sift_hip_overlay_draw (sift_amd/csrc/image_io.cpp) is compared with this, pixel for pixel, also for boxes that leave the image."""
import math

import numpy as np

f32 = np.float32


def box_of(kp, subpixel):
    """(cx, cy, side) of main.cpp:61-66: `u16_t x = (p.loc.x * std::pow(2, p.octave)) / subpixel_divisor` is computed in double and
    stored in 16 bits (x86 cvttsd2si, low half); cv::Size(p.scale * 10, p.scale * 10) holds ints (float -> int truncation)."""
    div = 2.0 if subpixel else 1.0
    cx = int((float(kp["x"]) * math.pow(2.0, float(kp["octave"]))) / div) & 0xFFFF
    cy = int((float(kp["y"]) * math.pow(2.0, float(kp["octave"]))) / div) & 0xFFFF
    side = int(f32(kp["scale"]) * f32(10))
    return cx, cy, side


def rotated_rect_points(cx, cy, w, h, angle):
    """cv::RotatedRect::points (OpenCV 3.2): b = (float)cos(a) * 0.5f, a = (float)sin(a) * 0.5f, every product and sum in float."""
    ang = float(f32(angle)) * math.pi / 180.0
    if math.isnan(ang) or math.isinf(ang):
        return None
    b = f32(f32(math.cos(ang)) * f32(0.5))
    a = f32(f32(math.sin(ang)) * f32(0.5))
    cx, cy, w, h = f32(cx), f32(cy), f32(w), f32(h)
    p0 = (f32(f32(cx - f32(a * h)) - f32(b * w)), f32(f32(cy + f32(b * h)) - f32(a * w)))
    p1 = (f32(f32(cx + f32(a * h)) - f32(b * w)), f32(f32(cy - f32(b * h)) - f32(a * w)))
    p2 = (f32(f32(f32(2) * cx) - p0[0]), f32(f32(f32(2) * cy) - p0[1]))
    p3 = (f32(f32(f32(2) * cx) - p1[0]), f32(f32(f32(2) * cy) - p1[1]))
    return [p0, p1, p2, p3]


def cv_round(v):
    """cvRound(float) = lrint: nearest, ties to even (Python's round on the exactly represented value does the same)."""
    return int(round(float(v)))


def trunc_div(num, den):
    """(int64)((double)num / den): the quotient truncated toward zero."""
    return int(float(num) / float(den))


def clip_line(width, height, x1, y1, x2, y2):
    """cv::clipLine(Size2l, Point2l&, Point2l&): Cohen-Sutherland, vertical borders after horizontal ones, every crossing truncated
    toward zero.  Returns None when nothing of the segment is inside."""
    if width <= 0 or height <= 0:
        return None
    right, bottom = width - 1, height - 1

    def code(x, y):
        return (x < 0) + (x > right) * 2 + (y < 0) * 4 + (y > bottom) * 8
    c1, c2 = code(x1, y1), code(x2, y2)
    if (c1 & c2) == 0 and (c1 | c2) != 0:
        if c1 & 12:
            a = 0 if c1 < 8 else bottom
            x1 += trunc_div((a - y1) * (x2 - x1), (y2 - y1))
            y1 = a
            c1 = (x1 < 0) + (x1 > right) * 2
        if c2 & 12:
            a = 0 if c2 < 8 else bottom
            x2 += trunc_div((a - y2) * (x2 - x1), (y2 - y1))
            y2 = a
            c2 = (x2 < 0) + (x2 > right) * 2
        if (c1 & c2) == 0 and (c1 | c2) != 0:
            if c1:
                a = 0 if c1 == 1 else right
                y1 += trunc_div((a - x1) * (y2 - y1), (x2 - x1))
                x1 = a
                c1 = 0
            if c2:
                a = 0 if c2 == 1 else right
                y2 += trunc_div((a - x2) * (y2 - y1), (x2 - x1))
                x2 = a
                c2 = 0
    return (x1, y1, x2, y2) if (c1 | c2) == 0 else None


def line_pixels(width, height, x1, y1, x2, y2):
    """The pixels cv::line sets: LineIterator(img, pt1, pt2, 8, leftToRight = true) - clipped first if an end point lies outside."""
    if not (0 <= x1 < width and 0 <= x2 < width and 0 <= y1 < height and 0 <= y2 < height):
        c = clip_line(width, height, x1, y1, x2, y2)
        if c is None:
            return []
        x1, y1, x2, y2 = c
    dx, dy = x2 - x1, y2 - y1
    if dx < 0:                      # left to right: start at the end point with the smaller x
        dx, dy, x1, y1 = -dx, -dy, x2, y2
    ystep = 1 if dy >= 0 else -1
    dy = abs(dy)
    steep = dy > dx
    major, minor = (dy, dx) if steep else (dx, dy)
    err = major - 2 * minor
    out, x, y = [], x1, y1
    for _ in range(major + 1):
        out.append((x, y))
        neg = err < 0
        err += -2 * minor + (2 * major if neg else 0)
        if steep:
            y += ystep
            x += 1 if neg else 0
        else:
            x += 1
            y += ystep if neg else 0
    return out


def draw_overlay(bgr, kps, subpixel):
    """main.cpp:60-73 on a copy of the B,G,R image `bgr` [h, w, 3] for the keypoint records `kps` (fields x, y, octave, scale,
    orientation)."""
    img = bgr.copy()
    h, w, _ = img.shape
    for kp in kps:
        cx, cy, side = box_of(kp, subpixel)
        pts = rotated_rect_points(cx, cy, side, side, kp["orientation"])
        if pts is None:
            continue
        # Point2f -> Point: saturate_cast<int>(float) = cvRound; a coordinate that is no finite int gives INT_MIN in OpenCV: nothing inside
        try:
            q = [(cv_round(x), cv_round(y)) for x, y in pts]
        except (ValueError, OverflowError):
            continue
        for a, b in ((0, 1), (0, 3), (2, 3), (1, 2)):
            for x, y in line_pixels(w, h, q[a][0], q[a][1], q[b][0], q[b][1]):
                img[y, x] = (255, 0, 0)
    return img

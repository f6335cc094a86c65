"""Constant-only stand-in for opencv (absent here); the fixture generator never decodes images."""
IMREAD_COLOR = 1
IMREAD_GRAYSCALE = 0
IMREAD_UNCHANGED = -1
COLOR_BGR2RGB = 4
COLOR_RGB2BGR = 4
COLOR_BGR2GRAY = 6
COLOR_GRAY2BGR = 8
COLOR_BGR2YCrCb = 36
BORDER_REFLECT_101 = 4
BORDER_REFLECT = 2
IMWRITE_JPEG_QUALITY = 1
IMWRITE_PNG_COMPRESSION = 16
INTER_LINEAR = 1
INTER_CUBIC = 2
INTER_AREA = 3
INTER_LANCZOS4 = 4


def __getattr__(name):  # any function call is a bug in the generator
    raise AttributeError(f"cv2 shim has no attribute {name}")

"""Minimal stand-in for pygame 2.1.2 -- FIXTURE GENERATION ONLY.

This is build-authored code, not reference code and not pygame.  pygame (the
reference pins pygame==2.1.2, /root/reference/requirements.txt:6) is absent
from this image and cannot be installed offline, so `make_golden.py` puts this
directory on sys.path to let the *unmodified* reference modules
(/root/reference/envs/battle_env.py, envs/sprites.py) import and run headless.

Only the part of pygame that carries arithmetic on the step() path is
restated with care -- `Rect` (integer x/y/w/h, truncating float->int
conversion, centre/edge accessors, strict-overlap colliderect), following the
published behaviour of pygame 2.1.2's src_c/rect.c / src_c/base.c:

  * pg_IntFromObj: a Python float (or subclass, e.g. numpy.float64) is
    converted with a C `(int)double` cast, i.e. truncation toward zero;
    anything else goes through PyLong_AsLong (needs __index__ on py>=3.10).
  * center setter:  x = cx - (w >> 1);  y = cy - (h >> 1)
    center getter:  (x + (w >> 1), y + (h >> 1))
  * right = x + w, bottom = y + h; the edge setters translate the rect.
  * colliderect: zero-size rects never collide; otherwise strict overlap
    A.left < B.right and A.top < B.bottom and A.right > B.left and
    A.bottom > B.top.  An argument that is not rect-like but has a `.rect`
    attribute is resolved through it.

Everything else (Surface, image, transform, sprite, display, font ...) is an
inert container: nothing they would compute reaches step()'s outputs.
"""
import math as _math
import operator as _operator
import struct as _struct
import types as _types

SRCALPHA = 0x00010000


def init():
    return (0, 0)


def quit():
    return None


def _to_int(v):
    # pg_IntFromObj (pygame 2.1.2 src_c/base.c)
    if isinstance(v, float):
        return int(v)  # C (int) cast: truncation toward zero
    return _operator.index(v)


def _two_ints(v):
    a, b = v
    return _to_int(a), _to_int(b)


class Rect:
    __slots__ = ("x", "y", "w", "h")

    def __init__(self, *args):
        if len(args) == 1:
            a = args[0]
            if isinstance(a, Rect):
                args = (a.x, a.y, a.w, a.h)
            else:
                args = tuple(a)
        if len(args) == 2:
            (x, y), (w, h) = args
        else:
            x, y, w, h = args
        self.x, self.y, self.w, self.h = _to_int(x), _to_int(y), _to_int(w), _to_int(h)

    # --- size
    @property
    def width(self):
        return self.w

    @property
    def height(self):
        return self.h

    @property
    def size(self):
        return (self.w, self.h)

    # --- edges
    @property
    def left(self):
        return self.x

    @left.setter
    def left(self, v):
        self.x = _to_int(v)

    @property
    def top(self):
        return self.y

    @top.setter
    def top(self, v):
        self.y = _to_int(v)

    @property
    def right(self):
        return self.x + self.w

    @right.setter
    def right(self, v):
        self.x = _to_int(v) - self.w

    @property
    def bottom(self):
        return self.y + self.h

    @bottom.setter
    def bottom(self, v):
        self.y = _to_int(v) - self.h

    @property
    def topleft(self):
        return (self.x, self.y)

    @topleft.setter
    def topleft(self, v):
        self.x, self.y = _two_ints(v)

    # --- centre
    @property
    def centerx(self):
        return self.x + (self.w >> 1)

    @centerx.setter
    def centerx(self, v):
        self.x = _to_int(v) - (self.w >> 1)

    @property
    def centery(self):
        return self.y + (self.h >> 1)

    @centery.setter
    def centery(self, v):
        self.y = _to_int(v) - (self.h >> 1)

    @property
    def center(self):
        return (self.x + (self.w >> 1), self.y + (self.h >> 1))

    @center.setter
    def center(self, v):
        cx, cy = _two_ints(v)
        self.x = cx - (self.w >> 1)
        self.y = cy - (self.h >> 1)

    def copy(self):
        return Rect(self.x, self.y, self.w, self.h)

    def colliderect(self, other):
        if not isinstance(other, Rect):
            if hasattr(other, "rect"):
                other = other.rect
                if callable(other):
                    other = other()
            else:
                other = Rect(other)
        a, b = self, other
        if a.w == 0 or a.h == 0 or b.w == 0 or b.h == 0:
            return False
        return (min(a.x, a.x + a.w) < max(b.x, b.x + b.w)
                and min(a.y, a.y + a.h) < max(b.y, b.y + b.h)
                and max(a.x, a.x + a.w) > min(b.x, b.x + b.w)
                and max(a.y, a.y + a.h) > min(b.y, b.y + b.h))

    def __repr__(self):
        return f"<rect({self.x}, {self.y}, {self.w}, {self.h})>"


class Surface:
    def __init__(self, size, flags=0, *a, **k):
        self._size = (int(size[0]), int(size[1]))

    def get_size(self):
        return self._size

    def get_width(self):
        return self._size[0]

    def get_height(self):
        return self._size[1]

    def get_rect(self, **kwargs):
        r = Rect(0, 0, self._size[0], self._size[1])
        for k, v in kwargs.items():
            setattr(r, k, v)
        return r

    def fill(self, *a, **k):
        return None

    def blit(self, *a, **k):
        return None

    def convert_alpha(self):
        return self


def _png_size(path):
    with open(path, "rb") as f:
        head = f.read(24)
    if head[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError(f"not a PNG: {path}")
    return _struct.unpack(">II", head[16:24])


image = _types.ModuleType("pygame.image")
image.load = lambda path: Surface(_png_size(path))


def _rotate(surface, angle):
    # bounding box of the rotated image (its size never reaches step() outputs:
    # Rect's centre set/get round-trips exactly for any size)
    w, h = surface.get_size()
    r = _math.radians(angle)
    c, s = abs(_math.cos(r)), abs(_math.sin(r))
    return Surface((max(1, int(round(w * c + h * s))), max(1, int(round(w * s + h * c)))))


transform = _types.ModuleType("pygame.transform")
transform.rotate = _rotate
transform.scale = lambda surface, size: Surface(size)


class Vector2:
    def __init__(self, x=0.0, y=None):
        if y is None and not isinstance(x, (int, float)):
            x, y = x
        elif y is None:
            y = x
        self.x, self.y = float(x), float(y)

    def __sub__(self, o):
        ox, oy = o if not isinstance(o, Vector2) else (o.x, o.y)
        return Vector2(self.x - ox, self.y - oy)

    def __add__(self, o):
        ox, oy = o if not isinstance(o, Vector2) else (o.x, o.y)
        return Vector2(self.x + ox, self.y + oy)

    def rotate(self, angle):
        r = _math.radians(angle)
        c, s = _math.cos(r), _math.sin(r)
        return Vector2(self.x * c - self.y * s, self.x * s + self.y * c)

    def __iter__(self):
        yield self.x
        yield self.y


math = _types.ModuleType("pygame.math")
math.Vector2 = Vector2


class _Sprite:
    def __init__(self, *groups):
        pass

    def kill(self):
        pass


sprite = _types.ModuleType("pygame.sprite")
sprite.Sprite = _Sprite


def _headless(*a, **k):
    raise RuntimeError("pygame stand-in is headless: rendering is out of scope")


display = _types.ModuleType("pygame.display")
display.quit = lambda: None
display.set_mode = _headless
display.update = _headless
font = _types.ModuleType("pygame.font")
font.Font = _headless
font.get_default_font = _headless
time = _types.ModuleType("pygame.time")
time.wait = _headless
time.Clock = _headless
draw = _types.ModuleType("pygame.draw")
draw.rect = _headless
event = _types.ModuleType("pygame.event")
event.get = lambda: []
surfarray = _types.ModuleType("pygame.surfarray")
surfarray.pixels3d = _headless

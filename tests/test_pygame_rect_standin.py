"""What the build ASSUMES about pygame.Rect (pygame 2.1.2, src_c/rect.c and src_c/base.c; third-party, absent from /root/reference and
not installable offline -- SURVEY.md section 8c), written down as a table of (Rect, operation) -> result.

The golden fixtures are produced by the UNMODIFIED reference running on build-authored stand-ins for the packages it imports; of those,
only `pygame.Rect` carries arithmetic that reaches the step() path (sprites.py:84-141 centre / edge stores, :312,:333 bullet rect,
:344,:349 colliderect).  pygame itself cannot be run here, so this link of the parity chain is pinned on what pygame documents and its own
test suite asserts, restated from memory of pygame 2.1.2's test/rect_test.py (test_center, test_centerx, test_right, test_bottom,
test_colliderect and the float-argument cases) and docs/reST/ref/rect.rst -- the cases are pygame's, the file is not.  Each row is one
assumption; the stand-in must satisfy it, and the row says where the reference leans on it."""
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden", "standins"))
import pygame  # noqa: E402  (the stand-in: tests/golden/standins/pygame)

Rect = pygame.Rect
sys.path.remove(os.path.join(HERE, "golden", "standins"))
sys.modules.pop("pygame", None)                           # nothing else in the suite should see the stand-in as `pygame`


def _r1():
    return Rect(1, 2, 3, 4)


ATTRIBUTES = [
    # (what, expression on a fresh Rect(1, 2, 3, 4), expected)                          pygame: rect_test.py test_left ... test_center; rect.rst "virtual attributes"
    ("right = x + w", lambda r: r.right, 4),
    ("bottom = y + h", lambda r: r.bottom, 6),
    ("centerx = x + w // 2 (integer: 1 + 1)", lambda r: r.centerx, 2),
    ("centery = y + h // 2", lambda r: r.centery, 4),
    ("center is the pair", lambda r: r.center, (2, 4)),
    ("size", lambda r: r.size, (3, 4)),
    ("an odd width's centre sits on the lower pixel: Rect(0, 0, 3, 3).center", lambda r: Rect(0, 0, 3, 3).center, (1, 1)),
    ("the plane sprite: Rect(0, 0, 50, 48).center (sprites.py:84: centre x in [25, 1175] after the clamps)", lambda r: Rect(0, 0, 50, 48).center, (25, 24)),
    ("the bullet sprite: a 6 x 3 rect centred on (100, 50) starts at (97, 49) (sprites.py:306-312: x0 = cx - 3, y0 = cy - 1)",
     lambda r: (lambda b: (setattr(b, "center", (100, 50)), b.topleft)[1])(Rect(0, 0, 6, 3)), (97, 49)),
]

SETTERS = [
    # (what, mutation, attribute read afterwards, expected)                              pygame: test_right / test_bottom / test_center "moves the rect, keeps the size"
    ("right = 10 moves the rect: x = 10 - w", lambda r: setattr(r, "right", 10), lambda r: (r.x, r.w), (7, 3)),
    ("bottom = 10 moves the rect", lambda r: setattr(r, "bottom", 10), lambda r: (r.y, r.h), (6, 4)),
    ("left = 0", lambda r: setattr(r, "left", 0), lambda r: (r.x, r.right), (0, 3)),
    ("top = 0", lambda r: setattr(r, "top", 0), lambda r: (r.y, r.bottom), (0, 4)),
    ("center = (22, 34): topleft moves by the same amount, size kept", lambda r: setattr(r, "center", (22, 34)), lambda r: (r.topleft, r.size), ((21, 32), (3, 4))),
    ("center set then read round-trips for any size (sprites.py:194-201: get_pos() == rect.center)",
     lambda r: setattr(r, "center", (640, 333)), lambda r: r.center, (640, 333)),
    # floats: pg_IntFromObj converts with a C (int) cast = truncation toward ZERO, not floor, not round (base.c; rect_test's float cases
    # construct Rect(1.2, 2.9, ...) -> (1, 2, ...)).  sprites.py:131,333 store float64 positions this way: x = -0.5 -> 0 stays on the field
    ("a float coordinate is truncated toward zero: center = (10.9, 7.99)", lambda r: setattr(r, "center", (10.9, 7.99)), lambda r: r.center, (10, 7)),
    ("... also below zero: center = (-0.5, -1.9) -> (0, -1), not (-1, -2)", lambda r: setattr(r, "center", (-0.5, -1.9)), lambda r: r.center, (0, -1)),
    ("Rect(1.2, 2.9, 3.7, 4.1) == Rect(1, 2, 3, 4)", lambda r: None, lambda r: (lambda q: (q.x, q.y, q.w, q.h))(Rect(1.2, 2.9, 3.7, 4.1)), (1, 2, 3, 4)),
]

COLLISIONS = [
    # (other rect as a function of r1 = Rect(1, 2, 3, 4), expected r1.colliderect(other))     pygame: rect_test.py test_colliderect; rect.rst: "touching edges do not overlap"
    ("overlaps the top-left corner", lambda r: Rect(0, 0, 2, 3), True),
    ("touches only the corner point", lambda r: Rect(0, 0, 1, 2), False),
    ("starts exactly at right / bottom: edges touch, no overlap (strict inequalities)", lambda r: Rect(r.right, r.bottom, 2, 2), False),
    ("strictly inside", lambda r: Rect(r.left + 1, r.top + 1, r.width - 2, r.height - 2), True),
    ("strictly around", lambda r: Rect(r.left - 1, r.top - 1, r.width + 2, r.height + 2), True),
    ("itself", lambda r: Rect(r), True),
    ("a zero-size rect never collides, even inside", lambda r: Rect(r.left, r.top, 0, 0), False),
    ("zero width only", lambda r: Rect(r.centerx, r.centery, 0, 2), False),
    ("shares the right edge from outside", lambda r: Rect(r.right, r.top, 1, 1), False),
    ("one pixel in from the right edge", lambda r: Rect(r.right - 1, r.top, 1, 1), True),
]


@pytest.mark.parametrize("what,expr,want", ATTRIBUTES, ids=[a[0][:50] for a in ATTRIBUTES])
def test_rect_attributes(what, expr, want):
    assert expr(_r1()) == want, what


@pytest.mark.parametrize("what,mutate,read,want", SETTERS, ids=[a[0][:50] for a in SETTERS])
def test_rect_setters_and_float_truncation(what, mutate, read, want):
    r = _r1()
    mutate(r)
    assert read(r) == want, what


@pytest.mark.parametrize("what,other,want", COLLISIONS, ids=[a[0][:50] for a in COLLISIONS])
def test_rect_colliderect_is_strict_overlap(what, other, want):
    r = _r1()
    assert bool(r.colliderect(other(r))) is want, what
    assert bool(other(r).colliderect(r)) is want, what + " (symmetric)"


def test_colliderect_resolves_an_object_through_its_rect_attribute():
    """sprites.py:344 passes a Base SPRITE, not a Rect: pygame takes any object with a `rect` attribute (pgRect_FromObject)."""
    class Sprite:
        def __init__(self, rect):
            self.rect = rect
    r = _r1()
    assert r.colliderect(Sprite(Rect(0, 0, 2, 3))) and not r.colliderect(Sprite(Rect(0, 0, 1, 2)))


def test_the_hit_windows_the_kernels_use_follow_from_these_rules():
    """a 6 x 3 bullet rect centred on (bx, by) against the 62 x 62 base rect centred on (Bx, By) overlaps  <=>
    dx in [-33, 33] and dy in [-32, 31]; against the un-rotated 50 x 48 plane rect  <=>  dx in [-27, 27], dy in [-25, 24] --
    the constants in csrc/bsx_step_phase_bullets.inl and oracle/battlespace_ref.c, derived here from Rect itself."""
    def window(w, h):
        target = Rect(0, 0, w, h)
        target.center = (600, 400)
        xs, ys = [], []
        for d in range(-40, 41):
            b = Rect(0, 0, 6, 3); b.center = (600 + d, 400)
            if b.colliderect(target):
                xs.append(d)
            b = Rect(0, 0, 6, 3); b.center = (600, 400 + d)
            if b.colliderect(target):
                ys.append(d)
        return (min(xs), max(xs)), (min(ys), max(ys))
    assert window(62, 62) == ((-33, 33), (-32, 31))
    assert window(50, 48) == ((-27, 27), (-25, 24))

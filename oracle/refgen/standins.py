"""In-memory stand-ins for `pygame` and `gym` -- TEST INFRASTRUCTURE ONLY.

The reference (harman097/RoboRugby) needs pygame and gym at import time but its
physics only touches a handful of container/int-rect behaviours.  Neither
package is installed in the build container and there is no network, so the
golden-vector generator (gen_golden.py) injects these minimal modules into
``sys.modules`` *before* importing the reference from ``/root/reference``.

Behaviours encoded here (assumptions recorded in every fixture's metadata):
  * ``pygame.Rect(l, t, w, h)``: every argument is truncated toward zero like a
    C ``(int)`` cast; ``right = left + width``; ``colliderect`` is a strict
    overlap test.  (Used by RR_Ball.py:8-15 / RR_TrashyPhysics.py:76-85 and by
    reset's spawn rejection RR_EnvBase.py:155-200.)
  * ``pygame.sprite.Group`` iterates in insertion order.
  * ``gym.Env`` exposes class attributes ``action_space``/``observation_space``
    = None, ``metadata``, ``reward_range`` and the ``unwrapped`` property.
Nothing in the product (roborugby_amd/) imports this file.
"""
import sys
import types


# --------------------------------------------------------------------------- pygame
class Rect:
    def __init__(self, left, top, width, height):
        self.left = int(left)
        self.top = int(top)
        self.width = int(width)
        self.height = int(height)

    @property
    def right(self):
        return self.left + self.width

    @property
    def bottom(self):
        return self.top + self.height

    @property
    def center(self):
        return (self.left + self.width // 2, self.top + self.height // 2)

    def colliderect(self, o):
        return (self.left < o.left + o.width and self.top < o.top + o.height and
                self.left + self.width > o.left and self.top + self.height > o.top)

    def inflate(self, dx, dy):
        return Rect(self.left - dx / 2, self.top - dy / 2, self.width + dx, self.height + dy)

    def __repr__(self):
        return f"<rect({self.left}, {self.top}, {self.width}, {self.height})>"


class Surface:
    def __init__(self, size=(0, 0), *a, **k):
        self._w, self._h = int(size[0]), int(size[1])

    def fill(self, *a, **k):
        pass

    def set_colorkey(self, *a, **k):
        pass

    def convert(self, *a, **k):
        return self

    def blit(self, *a, **k):
        pass

    def get_width(self):
        return self._w

    def get_height(self):
        return self._h

    def get_rect(self, **kw):
        r = Rect(0, 0, self._w, self._h)
        if "center" in kw:
            cx, cy = kw["center"]
            r.left = int(cx - self._w / 2)
            r.top = int(cy - self._h / 2)
        if "left" in kw:
            r.left = int(kw["left"])
        if "top" in kw:
            r.top = int(kw["top"])
        return r


class Sprite:
    # re-callable: Robot.on_reset re-runs __init__ while groups still hold it
    def __init__(self, *groups):
        if not hasattr(self, "_groups"):
            self._groups = []

    def alive(self):
        return len(self._groups) > 0

    def kill(self):
        for g in list(self._groups):
            g.remove(self)


class Group:
    def __init__(self, *sprites):
        self._sprites = []
        for s in sprites:
            self.add(s)

    def add(self, *sprites):
        for s in sprites:
            if s not in self._sprites:
                self._sprites.append(s)
                if not hasattr(s, "_groups"):
                    s._groups = []
                s._groups.append(self)

    def remove(self, *sprites):
        for s in sprites:
            if s in self._sprites:
                self._sprites.remove(s)
                s._groups.remove(self)

    def sprites(self):
        return list(self._sprites)

    def __iter__(self):
        return iter(self.sprites())

    def __len__(self):
        return len(self._sprites)

    def __bool__(self):
        return len(self._sprites) > 0


def collide_rect(a, b):
    return a.rect.colliderect(b.rect)


def spritecollide(sprite, group, dokill, collided=None):
    fn = collided or collide_rect
    hit = [s for s in group if fn(sprite, s)]
    if dokill:
        for s in hit:
            s.kill()
    return hit


def _noop(*a, **k):
    return None


def _build_pygame():
    pg = types.ModuleType("pygame")
    pg.init = _noop
    pg.quit = _noop
    pg.RLEACCEL = 0x4000
    for i, k in enumerate("ikolwsad"):
        setattr(pg, "K_" + k, 97 + i)
    pg.Rect = Rect
    pg.Surface = Surface
    pg.surface = types.ModuleType("pygame.surface")
    pg.surface.Surface = Surface

    sprite = types.ModuleType("pygame.sprite")
    sprite.Sprite = Sprite
    sprite.Group = Group
    sprite.collide_rect = collide_rect
    sprite.spritecollide = spritecollide
    pg.sprite = sprite

    display = types.ModuleType("pygame.display")
    display.set_mode = lambda size, *a, **k: Surface(size)
    display.flip = _noop
    pg.display = display

    image = types.ModuleType("pygame.image")
    image.load = lambda path: Surface((40, 20))
    image.tostring = lambda surf, fmt: bytes(3 * surf.get_width() * surf.get_height())
    pg.image = image

    draw = types.ModuleType("pygame.draw")
    draw.circle = draw.polygon = draw.rect = draw.line = _noop
    pg.draw = draw

    class _Font:
        def render(self, *a, **k):
            return Surface((10, 10))

    font = types.ModuleType("pygame.font")
    font.init = _noop
    font.SysFont = lambda *a, **k: _Font()
    pg.font = font

    transform = types.ModuleType("pygame.transform")
    transform.rotate = lambda surf, ang: surf
    pg.transform = transform

    mods = {"pygame": pg, "pygame.sprite": sprite, "pygame.display": display, "pygame.image": image,
            "pygame.draw": draw, "pygame.font": font, "pygame.transform": transform,
            "pygame.surface": pg.surface}
    return mods


# --------------------------------------------------------------------------- gym
class _Space:
    pass


class Box(_Space):
    def __init__(self, low, high, shape=None, dtype=None):
        import numpy as np
        if shape is None:
            low = np.asarray(low)
            shape = low.shape
        self.low = np.full(shape, low, dtype=dtype) if np.isscalar(low) else np.asarray(low, dtype=dtype)
        self.high = np.full(shape, high, dtype=dtype) if np.isscalar(high) else np.asarray(high, dtype=dtype)
        self.shape = tuple(shape)
        self.dtype = dtype


class Discrete(_Space):
    def __init__(self, n):
        self.n = n
        self.shape = ()


class Env:
    metadata = {}
    reward_range = (-float("inf"), float("inf"))
    spec = None
    action_space = None
    observation_space = None

    @property
    def unwrapped(self):
        return self


def _build_gym():
    import numpy as np
    gym = types.ModuleType("gym")
    gym.Env = Env
    spaces = types.ModuleType("gym.spaces")
    spaces.Box = Box
    spaces.Discrete = Discrete
    gym.spaces = spaces
    utils = types.ModuleType("gym.utils")
    seeding = types.ModuleType("gym.utils.seeding")
    seeding.np_random = lambda seed=None: (np.random.RandomState(seed), seed)
    utils.seeding = seeding
    gym.utils = utils
    envs = types.ModuleType("gym.envs")
    registration = types.ModuleType("gym.envs.registration")
    registration.registry = {}
    registration.register = lambda id, **kw: registration.registry.__setitem__(id, kw)
    envs.registration = registration
    gym.envs = envs
    return {"gym": gym, "gym.spaces": spaces, "gym.utils": utils, "gym.utils.seeding": seeding,
            "gym.envs": envs, "gym.envs.registration": registration}


def install():
    """Inject the stand-ins into sys.modules (idempotent)."""
    if "pygame" not in sys.modules:
        sys.modules.update(_build_pygame())
    if "gym" not in sys.modules:
        sys.modules.update(_build_gym())

"""vidmaker stand-in (fixture generation only): video export is out of scope and never called headless."""


class Video:
    """evaluate.py:78 calls env.start_recording() before its 10 recorded games; the fixture run plays none of them, so the
    object is created and never used."""

    def __init__(self, path, fps=None, resolution=None, **kw):
        self.path, self.fps, self.resolution = path, fps, resolution

    def update(self, *a, **k):
        pass

    def export(self, *a, **k):
        pass

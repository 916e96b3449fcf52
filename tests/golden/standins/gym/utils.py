"""gym.utils stand-in: EzPickle as a no-op mixin."""


class EzPickle:
    def __init__(self, *args, **kwargs):
        pass

"""PettingZoo 1.19.0 stand-in -- FIXTURE GENERATION ONLY (build-authored; pettingzoo is absent from this image)."""


class ParallelEnv:
    pass

def CaptureStdoutWrapper(env):
    return env


def AssertOutOfBoundsWrapper(env):
    return env


def OrderEnforcingWrapper(env):
    return env

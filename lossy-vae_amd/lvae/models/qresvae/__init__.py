from .zoo import *  # noqa: F401,F403

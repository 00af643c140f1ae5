from vapor_amd.prep import *  # noqa: F401,F403

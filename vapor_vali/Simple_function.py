from vapor_amd.simple_function import *  # noqa: F401,F403
from vapor_amd.simple_function import (invert_base, default_flank_length, default_read_length,  # noqa: F401
                                       default_max_sv_test)

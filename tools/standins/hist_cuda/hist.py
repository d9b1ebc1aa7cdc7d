"""CPU stand-in for hist_cuda.hist.hist (hist.py:39-51): the vote of
hist_cuda_core.cuh:40-60 restated in oracle/oracle_core.c."""
from oracle import core as _core


def hist(X, Y, min_x, min_y, min_z, max_x, max_y, max_z, len_x, len_y, len_z, mini_batch=8):
    return _core.hist_vote(X, Y, (min_x, min_y, min_z), (max_x, max_y, max_z),
                           (len_x, len_y, len_z))

import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from visma_amd import _lib, synth
src, tgt, T_gt, r = synth.make_pair(262144, 4194304, motion="radius")
c = _lib.Context(0); c.set_clouds_f64(src, tgt); c.set_nn_mode(_lib.NN_GRID)
c.iterate(np.eye(4), r, 6)
c.close()

import sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
from madrona_amd.simlib import Simulator, hip_lib_path
from parity_utils import run_pair
try:
    probs, step = run_pair("escape_room", 1, 120, flags=5, check_init=False)
    print('done', step, probs[:2])
except Exception as e:
    print('ERR', e)
with Simulator(hip_lib_path('escape_room'), 1, flags=5) as s:
    for i in range(120):
        s.step(1)
        try:
            d = s.dump_all()
        except Exception as e:
            print('step', i, 'ERR', e); break

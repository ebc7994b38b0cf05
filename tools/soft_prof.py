"""One soft-decision run of 17 superframes of 8k QAM64 7/8 (clean loopback) for rocprofv3: `rocprofv3 --kernel-trace --stats -d DIR -- python tools/soft_prof.py`."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import pyoracle as po
import gr_dvbt_amd as g

nsf = int(sys.argv[1]) if len(sys.argv) > 1 else 17
c = po.cfg(g.QAM64, g.C7_8, g.T8k)
iq = po.stream_slice(c, nsf, 21)
dev = torch.from_numpy(iq.view(np.float32)).cuda()
rx = g.Rx(g.QAM64, g.C7_8, g.T8k, max_samples=len(iq), soft_decision=1)
for _ in range(4):
    rx.run_device(dev.data_ptr(), len(iq))
torch.cuda.synchronize()
rx.close()

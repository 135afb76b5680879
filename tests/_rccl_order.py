"""Helper of test_gpu_sharded_planner.py: which HIP / HSA / RCCL libraries a process ends up with, by import order, and
whether a one-rank communicator comes up.  usage: python tests/_rccl_order.py drp_first|torch_first|no_torch"""
import faulthandler
import os
import sys
faulthandler.dump_traceback_later(45, exit=False)      # the parent gives up after 60 s: leave it every thread's stack first
sys.stdout.reconfigure(line_buffering=True)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
order = sys.argv[1]


def maps():
    out = set()
    for l in open('/proc/self/maps'):
        for k in ('libamdhip64', 'libhsa-runtime64', 'librccl'):
            if k in l:
                out.add(l.split()[-1])
    return sorted(out)


if order == 'torch_first':
    import torch  # noqa: F401
from dyn_res_pile_manip_amd.engine import Engine
e = Engine(0)
print('after Engine:', maps())
if order == 'drp_first':
    import torch  # noqa: F401
    print('after torch:', maps())
try:
    uid = e.comm_unique_id()
    e.comm_init(uid, 0, 1)
    print('communicator up:', e.comm_info())
except Exception as ex:
    print('ERR', ex)
print('at the end:', maps())

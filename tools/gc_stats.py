"""Candidate counts of k_graph_cells from a diagnostic build:
  hipcc ... -DGC_STATS -o ab/libdrp_gcstats.so
  DRP_LIB=ab/libdrp_gcstats.so python tools/gc_stats.py --particles 1200 --samples 512 --horizon 20 --steps 2 --warmup 1
(any bench.py arguments).  Prints, per sweep (A = first halo, B = the ring, 2 = emission): chunks of 16 per wave,
candidates and runs per quarter, and how many waves / quarters ran stage B."""
import ctypes, os, sys, io, contextlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dyn_res_pile_manip_amd import _lib
import bench

sys.argv = ['bench.py', '--no-alt', '--no-cpu-baseline'] + sys.argv[1:]
with contextlib.redirect_stdout(io.StringIO()):
    bench.main()
lib = ctypes.CDLL(_lib.LIB_PATH)
out = (ctypes.c_ulonglong * 16)()
assert lib.drp_gc_stats(out, 0) == 0
s = list(out)
waves, quarters = s[12], s[13]
print('waves %d quarters %d' % (waves, quarters))
for name, k in (('A', 0), ('B', 3), ('2', 6)):
    print('sweep %s: chunks/wave %.2f  candidates/quarter %.1f  runs/quarter %.2f' %
          (name, s[k] / waves, s[k + 1] / quarters, s[k + 2] / quarters))
print('stage B: %.1f %% of waves, %.1f %% of quarters' % (100 * s[14] / waves, 100 * s[15] / quarters))

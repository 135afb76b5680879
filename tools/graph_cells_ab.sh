cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_graph_strips.py -m gpu -q 2>&1 | tail -3
run() { python bench.py --particles $1 --samples $2 --horizon $3 --no-alt --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('N=$1 $4: %.3f ms  graph %.3f' % (d['ms_per_step'], d['kernel_ms_per_iteration']['graph']))"; }
for n in "300 1024 10" "450 1024 10" "600 1024 10" "1200 512 20"; do set -- $n
  DRP_NO_GRAPH_CELLS=1 run $1 $2 $3 strips
  DRP_GRAPH_CELLS_MIN_N=1 run $1 $2 $3 cells
done
for hb in 0.025 0.035 0.046 0.06; do for halo in 0.02 0.027 0.035; do DRP_GRAPH_CELLS_HB=$hb DRP_GRAPH_CELLS_HALO=$halo run 1200 512 20 "hb=$hb halo=$halo"; done; done
for hb in 0.06 0.09 0.13; do for halo in 0.04 0.055 0.07; do DRP_GRAPH_CELLS_MIN_N=1 DRP_GRAPH_CELLS_HB=$hb DRP_GRAPH_CELLS_HALO=$halo run 300 1024 10 "hb=$hb halo=$halo"; done; done

# scan of the chunk length T of the Lorenz-63 sweep lane kernel (and of extra compiler flags): tools/lane_t_scan.sh "flags" T...
set -e
FLAGS=$1; shift
touch vgpa_amd/csrc/ode_small.hip
VGPA_EXTRA_CFLAGS="-DVGPA_LANE_T_EXPERIMENTS $FLAGS" python -m vgpa_amd.build > gpurun_out/lane_t_build.log 2>&1
for t in "$@"; do
  VGPA_LANE_T_BWD=$t python tools/bench_small_configs.py 65536 L63 20 | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('flags [$FLAGS] T_BWD=$t', round(d['ms_per_step'],3), round(d['frac_of_8TBs'],3), d['phase_ms'])"
done

# usage: bash tools/trace_lde.sh [tag]: timeline of one batch of the D > 64 energy terms (bench.py's config4 block, D = 1024, 33 grid points),
# as built and with the round-4 schedule (VGPA_LDE_TWO_STREAMS=0 VGPA_LDE_INVERSE=rows VGPA_LDE_DIAG=valu VGPA_LDE_TILE_MAP=0 VGPA_LDE_SYRK_MIRROR=0 VGPA_LDE_PANEL=1 VGPA_LDE_K_DOWN=0) -> gpurun_out/<tag>_lde_trace_D1024_{new,old}.txt
export TMPDIR=/tmp
TAG=${1:-r05}
export VGPA_HEAD=${VGPA_HEAD:-$(cat vgpa_amd/_tree.txt 2>/dev/null)}
for v in new old; do
  rm -rf gpurun_out/tr_$v
  if [ $v = old ]; then export VGPA_LDE_TWO_STREAMS=0 VGPA_LDE_INVERSE=rows VGPA_LDE_DIAG=valu VGPA_LDE_TILE_MAP=0 VGPA_LDE_SYRK_MIRROR=0 VGPA_LDE_PANEL=1 VGPA_LDE_K_DOWN=0; fi
  rocprofv3 --kernel-trace -d gpurun_out/tr_$v -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-single-problem --no-config2 --no-config5 > gpurun_out/tr_$v.json 2> gpurun_out/tr_$v.err
  DB=$(find gpurun_out/tr_$v -name "*results.db" | head -1)
  { echo "# tree $VGPA_HEAD ($v)"; python3 tools/trace_lde.py $DB -v; } > gpurun_out/${TAG}_lde_trace_D1024_$v.txt
  rm -rf gpurun_out/tr_$v
done
head -22 gpurun_out/${TAG}_lde_trace_D1024_new.txt

#!/bin/bash
# Time, clock and package power of the forward / backward cover stepper ALONE (512 problems, launches back to back for 6 s) for the
# shipped build and the diagnostic ablations of ode_sym_impl.h (wrong results, timing / power only).
# Writes gpurun_out/<tag>_stepper_power_ablations.txt
TAG=${1:-r03}
out=gpurun_out/${TAG}_stepper_power_ablations.txt
: > $out
# usage: tools/power_ablations.sh [tag] ["name:flags" ...]   (default: the shipped build and the four ablations)
shift
if [ $# -eq 0 ]; then set -- "shipped:" "nofrag:-DVGPA_EXPERIMENTS -DVGPA_ABL_NOFRAG=1" "nostore:-DVGPA_EXPERIMENTS -DVGPA_ABL_NOSTORE=1" "novec:-DVGPA_EXPERIMENTS -DVGPA_ABL_NOVEC=1" "loop0:-DVGPA_SYM_LOOP1=0"; fi
names=""
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}; names="$names $name"
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off $flags -Ivgpa_amd/csrc -Iinclude tools/ubench/ode_sym_loop.hip -o gpurun_out/ode_sym_loop_$name 2> gpurun_out/ode_sym_loop_$name.err &
done
wait
for name in $names; do
  for fwd in 1 0; do
    gpurun_out/ode_sym_loop_$name 512 $fwd 6 > gpurun_out/ode_sym_loop.log &
    pid=$!
    sleep 3
    smp=""
    for i in 1 2 3 4; do smp="$smp $(rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|Power \(W\)' | paste - - | awk '{print $7 "/" $NF "W"}')"; sleep 0.5; done
    wait $pid
    echo "$name: $(cat gpurun_out/ode_sym_loop.log) |$smp" >> $out
  done
  rm -f gpurun_out/ode_sym_loop_$name
done
rm -f gpurun_out/ode_sym_loop.log
cat $out

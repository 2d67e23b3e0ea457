#!/bin/bash
# Writes the commit (and a +dirty mark) of the working tree into vgpa_amd/_tree.txt (git-ignored; it travels with the gpurun snapshot, the
# GPU box has no .git): the profile scripts put it into the header of every summary they write.  Run before a profiling gpurun call.
cd "$(dirname "$0")/.."
h=$(git rev-parse --short=12 HEAD)
git diff --quiet HEAD -- vgpa_amd include bench.py tools || h="$h+dirty"
echo $h > vgpa_amd/_tree.txt
echo $h

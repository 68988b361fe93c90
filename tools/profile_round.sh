#!/bin/bash
# The round's profile set, all from the SAME bench command (the launch configurations come from denet_amd/tuned/gfx950.json,
# so every pass runs the same kernels):  bash tools/profile_round.sh r02   (on the GPU box; results under gpurun_out/)
#   <tag>_kernel_stats.md   rocprofv3 --kernel-trace: per-kernel table of the roofline leg + of the timed steps
#   <tag>_pmc_traffic.json  FETCH_SIZE / WRITE_SIZE in two separate --pmc passes (HBM bytes per launch)
#   <tag>_pmc_mfma.json     SQ_VALU_MFMA_BUSY_CYCLES etc. in a third pass (MFMA utilisation per kernel)
TAG=${1:-r02}
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
CMD="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-warm --no-split-bf16 --no-configs --no-dp-selftest --no-h2d --no-instep --no-audit --no-inference"
O=gpurun_out/${TAG}_prof
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace -d $O/kt -o kt -- $CMD > $O/kt.log 2>&1
DB=$(find $O/kt -name "*.db" | head -1)
{ echo "# rocprofv3 --kernel-trace, $TAG"; echo; echo "command: \`rocprofv3 --kernel-trace -- $CMD\`"; echo; echo '```'; grep '^{' $O/kt.log | cut -c1-1200; echo '```'; echo;
  echo "## The 5 steps of the live roofline leg (every kernel alone on one stream)"; echo; python tools/rocpd_stats.py $DB --last-steps 5;
  echo; echo "## Warm-up excluded: the 10 timed steps + the 5 leg steps (two kernel chains overlap in the timed steps)"; echo; python tools/rocpd_stats.py $DB --skip-steps 3; } > gpurun_out/${TAG}_kernel_stats.md
# one timed step as a table (start, duration, queue, grid) and the timeline statistics of the timed steps (the 5 leg steps skipped)
python tools/step_trace.py $DB 8 > gpurun_out/${TAG}_step_trace.txt
python tools/timeline.py $DB 5 5 > gpurun_out/${TAG}_timeline.txt
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/f -o f -- $CMD > $O/f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/w -o w -- $CMD > $O/w.log 2>&1
python tools/pmc_traffic.py $(find $O/f -name "*.db" | head -1) $(find $O/w -name "*.db" | head -1) > gpurun_out/${TAG}_pmc_traffic.json
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $O/m -o m -- $CMD > $O/m.log 2>&1
python tools/pmc_mfma.py $(find $O/m -name "*.db" | head -1) > gpurun_out/${TAG}_pmc_mfma.json
python tools/hbm_table.py gpurun_out/${TAG} > gpurun_out/${TAG}_hbm_kernels.md
rm -rf $O
ls -la gpurun_out/${TAG}_*

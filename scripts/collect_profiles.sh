#!/bin/bash
# gpurun_out/prof_<tag>/ (scripts/profile_round6.sh <tag>) -> profiles/<prefix>_*: the summaries the round's documents cite. usage: collect_profiles.sh r06b r06
TAG=$1; PRE=$2; S=gpurun_out/prof_$TAG; D=profiles
for f in bench_line.json bench_line_driver_form.json bench_line_under_trace.json bench_no_overlap_staging.json bench_synchronous.json calibbench.txt frame_kernel_stats.txt \
         frame_timeline.txt framebench.txt framebench_raw.txt frontbench.txt gfbench.txt kernel_stats.txt kernel_stats_synchronous.txt lm_schedule_ab.txt segbench.txt step_timeline.txt \
         step_timeline_synchronous.txt thinbench.txt trackbench.txt residency.txt; do
  [ -f $S/$f ] && cp $S/$f $D/${PRE}_$f
done
{ for i in 1 2 3; do echo "--- pass $i: $(grep -m1 -o 'pmc [A-Za-z_ ]*' $S/pmc$i.log 2>/dev/null)"; cat $S/pmc$i.txt; done; } > $D/${PRE}_pmc_summary.txt
ls -la $D | grep ${PRE}_ | wc -l

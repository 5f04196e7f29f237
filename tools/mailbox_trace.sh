#!/bin/bash
# What a mailbox request's launches take ON THE DEVICE: rocprofv3 --kernel-trace --stats of the C driver (build/mailbox_driver, faithful mode,
# called form, the context's own RAMs), 40 requests of one size per run.  usage (GPU box): bash tools/mailbox_trace.sh <outdir> [sizes...]
set -u
out=${1:?output directory}; shift
sizes=${*:-"9 1024 4096 32767"}
export TMPDIR=/tmp
mkdir -p $out
for n in $sizes; do
  args=$(yes $n | head -40 | tr '\n' ' ')
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/n$n -- build/mailbox_driver $args > $out/n$n.log 2> $out/n$n.err
  echo "== NUM_PTS $n"
  cat "$(ls -t $out/n$n/*/*_kernel_stats.csv | head -1)"
done

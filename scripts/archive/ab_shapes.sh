#!/bin/bash
# A/B two builds (lib/liblsqrhip.so vs lib/liblsqrhip_head.so) on the gather-bound shapes
cd "$(dirname "$0")/.."
L=lsqr_amd/lib
cp $L/liblsqrhip.so /tmp/new.so
for spec in random:1250000:10000000:100 random:4000000:1000000:100 powerlaw:5000000:2000000:10000 poisson2d:1000:1000; do
  for v in new head; do
    if [ $v = new ]; then cp /tmp/new.so $L/liblsqrhip.so; else cp $L/liblsqrhip_head.so $L/liblsqrhip.so; fi
    echo -n "$v  "; timeout 300 python scripts/kernel_times.py $spec 40 2>/dev/null
  done
done
cp /tmp/new.so $L/liblsqrhip.so

#!/bin/bash
# A/B two builds of liblsqrhip.so on one box with any script: lib/liblsqrhip.so vs lib/liblsqrhip_head.so
cd "$(dirname "$0")/.."
L=lsqr_amd/lib
cp $L/liblsqrhip.so /tmp/new.so
for r in 1 2; do
  cp /tmp/new.so $L/liblsqrhip.so; echo "== new"; timeout 200 python "$@" 2>/dev/null
  cp $L/liblsqrhip_head.so $L/liblsqrhip.so; echo "== head"; timeout 200 python "$@" 2>/dev/null
done
cp /tmp/new.so $L/liblsqrhip.so

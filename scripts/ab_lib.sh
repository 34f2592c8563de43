#!/bin/bash
# A/B two builds of liblsqrhip.so on one box: lib/liblsqrhip.so vs lib/liblsqrhip_head.so
cd "$(dirname "$0")/.."
L=lsqr_amd/lib
cp $L/liblsqrhip.so /tmp/new.so
for r in 1 2; do
  for v in 1 0; do
    cp /tmp/new.so $L/liblsqrhip.so; echo "new VAL8=$v"; LSQRHIP_VAL8=$v timeout 100 python scripts/poll_cost.py 2>/dev/null | sed -n 2p
    cp $L/liblsqrhip_head.so $L/liblsqrhip.so; echo "head VAL8=$v"; LSQRHIP_VAL8=$v timeout 100 python scripts/poll_cost.py 2>/dev/null | sed -n 2p
  done
done
cp /tmp/new.so $L/liblsqrhip.so

bash scripts/r05_phased.sh
python3 scripts/ab_env.py random:10000000:10000000:100 LSQRHIP_CSB_NARROW=0,1 5 3 2>&1 | tail -2
python3 scripts/ab_env.py random:1250000:10000000:100 LSQRHIP_CSB_NARROW=0,1 5 3 2>&1 | tail -2
python3 scripts/ab_env.py powerlaw:5000000:2000000:10000 LSQRHIP_CSB_NARROW=0,1 5 3 2>&1 | tail -2

# headline step time for several EQ || convolution cut-point sets (OHS_OVERLAP_CUTS); arguments = sets, default a sweep
sets=("$@")
[ ${#sets[@]} -eq 0 ] && sets=("0.34,0.66,0.92" "0.25,0.5,0.75,0.92,0.98" "0.25,0.5,0.75,0.9,0.96,0.99" "0.2,0.4,0.6,0.8,0.92,0.98" "0.3,0.6,0.9,0.98" "0.25,0.5,0.75,0.93,0.985" "0.34,0.66,0.92" "0.25,0.5,0.75,0.92,0.98")
for c in "${sets[@]}"; do
  OHS_OVERLAP_CUTS=$c timeout -k 10 300 python3 bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$c', d['value'], d['ms_per_step'])"
done

# the native latency harness under different values of one environment variable:  bash tools/ab_rt_env.sh VAR v1 v2 ...
var=$1; shift
for v in "$@"; do
  echo "$var=$v"; env "$var=$v" timeout -k 10 120 tools/bin/rt_latency | grep "1024_resident_kernel\|256_resident\|8_frames"
done

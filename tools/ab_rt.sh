# A/B of library variants under the native latency harness: tools/bin/var_<tag>/libohs_hip.so
for v in base "$@"; do
  echo "VARIANT $v"
  if [ "$v" = base ]; then timeout -k 10 120 tools/bin/rt_latency | grep -B1 "1024_resident_kernel\|256_resident" | grep -v "^--\|device phases"
  else LD_LIBRARY_PATH=$PWD/tools/bin/var_$v timeout -k 10 120 tools/bin/rt_latency | grep -B1 "1024_resident_kernel\|256_resident" | grep -v "^--\|device phases"; fi
done

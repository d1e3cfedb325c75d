# in-process timing of k_conv_p1 for library variants: base, then open_headstage_amd/libohs_hip_<tag>.so, twice
# (every OHS_P1_TARGET_WAVES setting of tools/tune_p1.py is printed: the best one depends on the waves per CU)
for v in base "$@" base "$@"; do
  if [ "$v" = base ]; then lib=$PWD/open_headstage_amd/libohs_hip.so; else lib=$PWD/open_headstage_amd/libohs_hip_$v.so; fi
  echo "VARIANT $v"; OHS_LIB=$lib python3 tools/tune_p1.py 2>&1 | grep "TARGET_WAVES" | sed -e "s/'OHS_P1_TARGET_WAVES': //" -e "s/, 'OHS_P1_STAGGER': '0'//"
done

for v in "" ko_TW ko_CD ko_twcd ko_STORE ""; do
  if [ -z "$v" ]; then lib=$PWD/open_headstage_amd/libohs_hip.so; else lib=$PWD/open_headstage_amd/libohs_hip_$v.so; fi
  echo "VARIANT ${v:-base}"; OHS_LIB=$lib python3 tools/tune_p1.py 2>&1 | grep "7680.*STAGGER.: .0"
done

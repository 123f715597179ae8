python -m pytest tests/test_model_gpu.py -x -q 2>&1 | tail -2
for lib in - ; do
  echo "== $lib serialized"; AMD_SERIALIZE_KERNEL=3 KWS_AB_ROWS=50 python tools/benchab.py $lib --steps 30 --no-cpu-baseline --no-extra 2>&1 | grep -E "group|dgrad_clip|wgrad_clip|^-"
done
python tools/ab_libs.py 2 - tools/lib_prev.bin 2>&1 | cut -c1-60

mkdir -p gpurun_out/r7u
for v in default w64_NOMMA w64_NODMA w64_NOCHUNKEND w64_NOREAD w64_NODMA_NOCHUNKEND w64_ONLYDMA w64_ONLYMMA; do
  if [ $v = default ]; then unset DL_LIB_PATH; else export DL_LIB_PATH=variants/libdisenlink_hip_$v.so; fi
  timeout -k 10 120 python tools/project_fwd_quick.py 5201 128 8 512 64 5201 2088 8 512 64 >> gpurun_out/r7u/bounds.txt 2>&1 || exit 1
done
unset DL_LIB_PATH
DL_PROJ_EIGHT_WAVES=1 timeout -k 10 120 python tools/project_fwd_quick.py 5201 128 8 512 64 5201 2088 8 512 64 >> gpurun_out/r7u/bounds.txt 2>&1
cat gpurun_out/r7u/bounds.txt

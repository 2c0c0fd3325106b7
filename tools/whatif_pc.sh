mkdir -p gpurun_out/r2c
for d in ${PC_DEBUG_LIST:-0 7 15}; do
  echo "== CY_PC_DEBUG=$d" 
  CY_PC_DEBUG=$d timeout -k 10 120 python tools/diag_pc.py ${PC_N:-32} time ${PC_LAYERS:-Conv1b,Up3,Up4,Up_conv5b,Up2} 2>&1 | grep -v amdgpu
done

# The packed-fp32 victims of pk_mfma_hazard.hip in one process, the real U-Net in another, on the same GPU.
#   usage: bash scripts/probe/pk_hazard_xproc.sh            (on the GPU box; needs the built probe binary)
cd $GRAFT_REPO_ROOT
unet_loop() {   # $1 = CT_CONV_MATH
rm -f /tmp/unet_loop.log
( CT_CONV_MATH=$1 timeout 90 python - <<'PY' > /tmp/unet_loop.log 2>&1
import importlib, torch, time
mod = lambda n: importlib.import_module("3deecelltracker_amd." + n)
synth, unet3d = mod("synth"), mod("unet3d")
model = unet3d.unet3_a().set_weights_dict(synth.make_unet_weights("unet3_a", 0))
vol = torch.randn(512, 512, 32, device="cuda"); out = torch.zeros_like(vol)
print("running", flush=True)
t0 = time.time()
while time.time() - t0 < 22:
    for _ in range(20): model.predict_volume_device(vol, out=out)
    torch.cuda.synchronize()
PY
) &
for i in $(seq 1 80); do grep -q running /tmp/unet_loop.log 2>/dev/null && break; sleep 1; done
sleep 1
}
echo "-- alone"; scripts/probe/pk_mfma_hazard victimonly 100; scripts/probe/pk_mfma_hazard victim3only 100
echo "-- beside the U-Net (split-fp16 conv kernels) of another process"; unet_loop f16x3
scripts/probe/pk_mfma_hazard victimonly 200; scripts/probe/pk_mfma_hazard victim1only 50; scripts/probe/pk_mfma_hazard victim3only 200
if [ -x scripts/probe/pk_mfma_hazard_wait0 ]; then echo "   (same victim built with -mllvm -amdgpu-waitcnt-forcezero=1: every wait is for ALL outstanding operations)"; scripts/probe/pk_mfma_hazard_wait0 victimonly 200; fi
scripts/probe/pk_mfma_hazard victimonly 100; wait
if [ "${1:-}" = f32 ]; then echo "-- beside the U-Net with the f32-input MFMA conv kernels"; unet_loop f32; scripts/probe/pk_mfma_hazard victimonly 300; wait; fi

# shader clock and package power while the U-Net runs back to back (rocm-smi sampled every 0.5 s)
python - <<'PY' &
import importlib, sys, time, torch
sys.path.insert(0, ".")
m = lambda n: importlib.import_module("3deecelltracker_amd." + n)
model = m("unet3d").unet3_a().set_weights_dict(m("synth").make_unet_weights("unet3_a", 0))
vol = torch.randn(512, 512, 32, device="cuda"); out = torch.zeros_like(vol)
t0 = time.time()
while time.time() - t0 < 8:
    for _ in range(20): model.predict_volume_device(vol, out=out)
    torch.cuda.synchronize()
PY
sleep 3
for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|mclk" | tr '\n' ' '; echo; sleep 0.5; done
wait
echo idle; sleep 2; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo

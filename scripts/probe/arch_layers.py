import sys, ctypes, importlib, torch
sys.path.insert(0, '.')
def mod(n): return importlib.import_module("3deecelltracker_amd." + n)
synth, unet3d, archs, _lib = mod("synth"), mod("unet3d"), mod("arch").ARCHS, mod("_lib")
batch = {"unet3_a": 75, "unet3_c": 150, "unet3_b": 24}
for name in sys.argv[1:]:
    arch = archs[name]; nb = batch[name]
    model = getattr(unet3d, name)().set_weights_dict(synth.make_unet_weights(name, 0))
    x = torch.randn(nb, *arch.input_shape, device="cuda")
    for _ in range(2): model.predict_device(x)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    for _ in range(5): model.predict_device(x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"{name}: {nb} patches {dt*1e3:.2f} ms  {nb*arch.flops_per_patch()/dt/1e12:.1f} TFLOP/s fp32-equivalent ({dt/nb*1e3:.3f} ms/patch)")
    L = _lib.lib(); h = model._handle
    L.ct_unet_set_timing(h, 1)
    for _ in range(3): model.predict_device(x)
    nl = L.ct_unet_num_conv_layers(h)
    ms = (ctypes.c_float * nl)(); cnt = (ctypes.c_int * nl)()
    L.ct_unet_get_timing(h, ms, cnt, nl); L.ct_unet_set_timing(h, 0)
    convs = arch.conv_layers()
    dims = None
    for i in range(nl):
        print(f"  L{i}: {ms[i]/3:.3f} ms ({cnt[i]//3} launches) cin,cout={convs[i] if i < len(convs) else 'head'}")

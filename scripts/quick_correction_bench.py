"""Accurate correction on a BASELINE-sized frame (512x512x32 prob map, 600 cells) -- dev helper."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
synth = importlib.import_module("3deecelltracker_amd.synth"); cit = importlib.import_module("3deecelltracker_amd.coord_image_transformer")
shape, f, n = (512, 512, 32), 5, 600
case = synth.make_correction_case(3, shape, f, n, 10)
vol1 = cit.Coordinates(case["vol1"], f, case["voxel_size"], "raw")
tr = cit.CoordsToImageTransformer(shape, case["voxel_size"], f, case["subregions"], vol1)
coords = cit.Coordinates(case["coords0"], f, case["voxel_size"], "raw")
prob_d = torch.from_numpy(case["prob"]).cuda()
for _ in range(2): out = tr.accurate_correction(prob_d, coords, ensemble=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): out = tr.accurate_correction(prob_d, coords, ensemble=True)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print(f"accurate_correction {shape} {n} cells: {dt*1e3:.2f} ms ({tr.last_iterations} rounds)")

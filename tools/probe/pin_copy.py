"""H2D rate of pinned sources made in different ways (development probe)."""
import time, torch
dev = torch.device("cuda", 0)
shape = (256, 3, 480, 480)
dst = torch.empty(shape, dtype=torch.uint8, device=dev)
frames = torch.randint(0, 256, shape, dtype=torch.uint8)
srcs = {"empty().pin_memory()": torch.empty(shape, dtype=torch.uint8).pin_memory(),
        "frames.pin_memory()": frames.pin_memory(),
        "frames.roll().pin_memory()": frames.roll(1, 0).pin_memory(),
        "empty(pin_memory=True)": torch.empty(shape, dtype=torch.uint8, pin_memory=True)}
srcs["empty().pin_memory() then copy_(frames)"] = torch.empty(shape, dtype=torch.uint8).pin_memory()
srcs["empty().pin_memory() then copy_(frames)"].copy_(frames)
st = torch.cuda.Stream(device=dev)
for name, src in srcs.items():
    for r in range(3):
        torch.cuda.synchronize(); a = time.perf_counter()
        with torch.cuda.stream(st):
            dst.copy_(src, non_blocking=True)
        b = time.perf_counter(); torch.cuda.synchronize(); c = time.perf_counter()
    print(f"{name:45s} is_pinned {src.is_pinned()} host {1e3*(b-a):7.3f} ms total {1e3*(c-a):7.3f} ms  {src.numel()/(c-a)/1e9:6.1f} GB/s")

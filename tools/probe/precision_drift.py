"""Loss trajectory of fused training steps in f32 vs bf16x3 conv arithmetic (same data, same dropout)."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import torch, fdet_amd, oracle as O
    from fdet_amd.models import ModelMeta
    from fdet_amd.models.PoolResnet import PoolResnet
    from fdet_amd import hotpath as hp
    torch.manual_seed(0)
    m = PoolResnet(64, (3, 480, 480), 10).cuda().train()
    mm = ModelMeta(model=m); mm.configure_optimizers()
    B = 64
    g = torch.Generator().manual_seed(100)
    x = torch.rand(B, 3, 480, 480, generator=g)
    if len(sys.argv) > 2 and sys.argv[2] == "perturb":
        x = x + 1e-7 * torch.rand(x.shape, generator=g)
    x = x.cuda()
    y = hp.encode_targets(O.synthetic_boxes(B, 480, seed=101), (480, 480), 10)
    out = []
    for s in range(14):
        l, _, _ = mm.fused_train_step(x, y)
        out.append(float(l))
    print(json.dumps(out))
else:
    for name, env, extra in (("f32", {"FDET_PRECISION": "f32"}, []), ("f32+1e-7 input noise", {"FDET_PRECISION": "f32"}, ["perturb"]), ("bf16x3", {"FDET_PRECISION": "bf16x3"}, [])):
        r = subprocess.run([sys.executable, __file__, "child"] + extra, env=dict(os.environ, **env), capture_output=True, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("[")]
        print(name, [round(v, 4) for v in json.loads(line[-1])] if line else r.stderr[-400:])

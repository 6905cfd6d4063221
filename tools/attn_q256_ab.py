import json, os, subprocess, sys
ROOT = "/root/repo" if os.path.exists("/root/repo/bench.py") else os.getcwd()
def worker(views, heads):
    import torch
    sys.path.insert(0, ROOT)
    from fast3r_amd import ops
    T = views * 1024
    g = torch.Generator(device="cuda").manual_seed(1)
    q = (torch.randn((T, heads * 64), generator=g, device="cuda") * 0.35).half()
    k = (torch.randn((T, heads * 64), generator=g, device="cuda") * 1.5).half()
    vt = torch.randn((heads * 64, T), generator=g, device="cuda").half()
    o = torch.empty_like(q)
    for _ in range(3):
        ops.attention(q, o, heads, 1.0, [(k, vt, T, 0, 0)], q_prescaled=True, kernel_sel=2)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    iters = 20 if views <= 20 else 5
    e0.record()
    for _ in range(iters):
        ops.attention(q, o, heads, 1.0, [(k, vt, T, 0, 0)], q_prescaled=True, kernel_sel=2)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(json.dumps({"ms": ms, "tflops": 4.0 * T * T * 64 * heads / ms / 1e9}))
if len(sys.argv) > 1 and sys.argv[1] == "worker":
    worker(int(sys.argv[2]), int(sys.argv[3])); sys.exit(0)
for views in (3, 8, 20, 40, 100):
    for rnd in range(2):
        for force in ("0", "1", ""):
            env = dict(os.environ); env.pop("F3R_ATTN_Q256", None)
            if force: env["F3R_ATTN_Q256"] = force
            p = subprocess.run([sys.executable, __file__, "worker", str(views), "16"], env=env, capture_output=True, text=True)
            line = [l for l in p.stdout.splitlines() if l.startswith("{")]
            rec = json.loads(line[-1]) if line else {"error": p.stderr[-200:]}
            rec.update(views=views, round=rnd, form={"0": "512-query items", "1": "256-query items", "": "library's choice"}[force])
            print(json.dumps(rec), flush=True)

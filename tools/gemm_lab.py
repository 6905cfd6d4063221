"""Measurement harness for VARIANTS of the hand-scheduled GEMM kernels (fast3r_amd/csrc/asm/gemm_gen.py): code objects built on the CPU box
(python tools/gemm_lab.py --build: one .hsaco per --ablate set under tools/lab/obj/, which travels with the tree) are loaded with
hipModuleLoad and launched directly -- no library in between -- on random data, interleaved rounds, HIP events on the launch stream.

    python tools/gemm_lab.py --build                      (CPU box)
    python tools/gemm_lab.py --run [--shapes ...]         (GPU box)
Prints one JSON line per (variant, role, shape).  Ablated variants compute garbage by construction (timing only)."""
import argparse
import ctypes
import json
import os
import subprocess
import sys

here = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(here, "..", "fast3r_amd", "csrc", "asm"))
import gemm_gen  # noqa: E402

OBJ = os.path.join(here, "lab", "obj")
LLVM = "/opt/rocm/lib/llvm/bin"
VARIANTS = {"product": [], "nodma": ["nodma"], "nolds": ["nolds"], "nobarrier": ["nobarrier"], "nodma_nolds": ["nodma", "nolds"],
            "mfma_only": ["nodma", "nolds", "nobarrier"]}


def build(extra=None):
    os.makedirs(OBJ, exist_ok=True)
    variants = dict(VARIANTS)
    for name, kw in (extra or {}).items():
        variants[name] = kw
    for name, abl in variants.items():
        kw = dict(ablate=abl) if isinstance(abl, list) else abl
        gens = gemm_gen.product_generators(**kw)
        s = os.path.join(OBJ, f"gemm_{name}.s")
        open(s, "w").write(gemm_gen.module_text(gens))
        subprocess.run([f"{LLVM}/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", s, "-o", s[:-2] + ".o"], check=True)
        subprocess.run([f"{LLVM}/ld.lld", "-shared", s[:-2] + ".o", "-o", s[:-2] + ".hsaco"], check=True)
        os.remove(s[:-2] + ".o")
        os.remove(s)   # (generated text: the .hsaco is what travels to the GPU box; neither is tracked)
        print("built", s[:-2] + ".hsaco")


class Hip:
    def __init__(self):
        self.l = ctypes.CDLL("libamdhip64.so")
        self.mods = {}

    def fn(self, path, name):
        if path not in self.mods:
            m = ctypes.c_void_p()
            assert self.l.hipModuleLoad(ctypes.byref(m), path.encode()) == 0, path
            self.mods[path] = m
        f = ctypes.c_void_p()
        assert self.l.hipModuleGetFunction(ctypes.byref(f), self.mods[path], name.encode()) == 0, name
        return f

    def launch(self, f, grid, karg, stream):
        buf = ctypes.create_string_buffer(karg, len(karg))
        size = ctypes.c_size_t(len(karg))
        cfg = (ctypes.c_void_p * 5)(1, ctypes.cast(buf, ctypes.c_void_p).value, 2, ctypes.cast(ctypes.byref(size), ctypes.c_void_p).value, 3)
        r = self.l.hipModuleLaunchKernel(f, grid, 1, 1, 256, 1, 1, 0, ctypes.c_void_p(stream), None, cfg)
        assert r == 0, r


def run(args):
    import torch
    hip = Hip()
    dev = "cuda"
    dt = torch.float16 if args.dtype == "f16" else torch.bfloat16
    shapes = []
    for sp in args.shapes.split(","):
        role, m, n, k, *rest = sp.split(":")
        shapes.append((role, int(m), int(n), int(k), rest))
    names = args.variants.split(",") if args.variants else sorted(f[5:-6] for f in os.listdir(OBJ) if f.endswith(".hsaco"))
    stream = torch.cuda.current_stream().cuda_stream
    for role, M, N, K, rest in shapes:
        segs = 2 if "w2" in rest else 1
        act = gemm_gen.ACT_GELU if "gelu" in rest else gemm_gen.ACT_NONE
        a = torch.randn((M, K), device=dev).to(dt)
        w = (torch.randn((N, K * segs), device=dev) * K ** -0.5).to(dt)
        bias = torch.randn(N, device=dev)
        esize = 4 if role == "f32" else 2
        out = torch.randn((M, N), device=dev) if role == "f32" else torch.empty((M, N), dtype=dt, device=dev)
        res = out.data_ptr() if (role == "f32" and "nores" not in rest) else 0
        karg, n_wg = gemm_gen.pack_args(a.data_ptr(), w.data_ptr(), 0 if "nobias" in rest else bias.data_ptr(), res, out.data_ptr(), K * 2, K * segs * 2, N * 4,
                                        N * esize, K // 64 * segs, K // 64, M // 256, N // 256, act)
        fns = {nm: hip.fn(os.path.join(OBJ, f"gemm_{nm}.hsaco"), f"f3r_gemm_asm_{role}_{args.dtype}") for nm in names}
        times = {nm: [] for nm in names}
        for nm in names:
            hip.launch(fns[nm], n_wg, karg, stream)
        torch.cuda.synchronize()
        for _ in range(args.rounds):
            for nm in names:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.inner):
                    hip.launch(fns[nm], n_wg, karg, stream)
                e1.record()
                torch.cuda.synchronize()
                times[nm].append(e0.elapsed_time(e1) / args.inner)
        for nm in names:
            ms = sorted(times[nm])[len(times[nm]) // 2]
            tiles_per_cu = (M // 256) * (N // 256) / 256.0
            print(json.dumps({"variant": nm, "role": role, "opts": rest, "dtype": args.dtype, "M": M, "N": N, "K": K, "segs": segs, "ms": round(ms, 4),
                              "tflops_mfma": round(2.0 * M * N * K * segs / ms / 1e9, 1), "us_per_tile_round": round(ms * 1e3 / tiles_per_cu, 2)}), flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--build", action="store_true")
    ap.add_argument("--run", action="store_true")
    ap.add_argument("--dtype", default="f16")
    ap.add_argument("--variants", default="")
    ap.add_argument("--shapes", default="lp:327680:4096:256,lp:327680:4096:1024,lp:327680:4096:4096,lp:327680:4096:1024:gelu,f32:327680:1024:1024,f32:327680:1024:4096,"
                                        "f32:327680:1024:1024:nores,lp:327680:4096:1024:w2")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--inner", type=int, default=3)
    a = ap.parse_args()
    if a.build:
        build()
    if a.run:
        run(a)

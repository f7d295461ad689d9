#!/usr/bin/env python3
"""A/B timing of tuning knobs on one workload, one process (developer tool).

    python scripts/ab.py --scale 20 --steps 3 base no_wmajor=1 window=16384 ...

Every positional argument is a comma-separated list of knob=value settings ("base" = defaults);
each variant runs on a fresh context.  Prints the stage times (HIP events) and the digest, which
must not change between variants."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=int, default=20)
    ap.add_argument("--workload", default="rmat")
    ap.add_argument("--grid", type=int, default=0)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--sink", default="digest")
    ap.add_argument("--flags", type=int, default=0, help="sink flags (2 ordered, 8 exact pattern)")
    ap.add_argument("--signed", action="store_true", help="flip the sign of half of the operand's values (cancellation: the EXACT_PATTERN paths)")
    ap.add_argument("variants", nargs="*", default=["base"])
    args = ap.parse_args()
    import torch
    import bench
    from spsparse_amd import capi
    dev = torch.device("cuda", 0)
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    ref = None
    for var in args.variants:
        ctx = capi.Context(0, stream.cuda_stream)
        if var != "base":
            for kv in var.split(","):
                k, v = kv.split("=")
                ctx.set_tuning(k, int(v))
        w = bench.Workload(torch, capi, ctx, dev, args.workload, args.scale, args.grid, 1, args.sink)
        w.flags = args.flags
        if args.signed:
            g = torch.Generator(device=dev); g.manual_seed(7)
            w.t[2].mul_(torch.where(torch.rand(w.t[2].numel(), device=dev, generator=g) < 0.5, -1.0, 1.0))
            torch.cuda.synchronize()
        out, results = bench.run_workload(torch, w, args.steps, 1)
        st = out["stage_ms"]
        dig = out.get("digest", {}).get("hash"), tuple(out["nnz_c"])
        if ref is None:
            ref = dig
        print("%-28s %8.2f ms/step  %s  %s" % (var, out["ms_per_step"],
              " | ".join("c %.2f s %.2f l %.2f m %.2f h %.2f d %.2f" % (x["consolidate"], x["symbolic"], x["light"], x["mid"],
                         x["heavy_hash_cells"], x["heavy_dense_cells"]) for x in st),
              "OK" if dig == ref else "DIGEST DIFFERS %s vs %s" % (dig, ref)), flush=True)
        w.release()
        ctx.close()
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()

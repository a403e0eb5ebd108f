#!/usr/bin/env python3
"""HBM a key and its context hold, per curve and key size (INTEGRATION.md section 5 quotes this table):
window table, direct-sum table, twin (after amsm_bases_prebuild_twin), and the context's grow-only workspace after a
batch of three MSMs of the key's length.   python tools/memory_table.py [--max-log2 22] > profiles/r06_memory_table.md"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def mib(b):
    return f"{b / (1 << 20):.0f}" if b >= (1 << 20) else f"{b / (1 << 20):.2f}"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--max-log2", type=int, default=22)
    a = ap.parse_args()
    from accumulation_amd import CommitterKey, Context, VariableBaseMSM, ffi
    print("| curve | generators | window bits / levels | window table MiB | direct-sum table MiB | twin MiB | plain key MiB | workspace after 3 MSMs MiB |")
    print("|---|---|---|---|---|---|---|---|")
    for curve, name in ((ffi.AMSM_PALLAS, "Pallas"), (ffi.AMSM_BLS12_381_G1, "BLS12-381 G1")):
        for lg in (10, 12, 14, 15, 16, 18, 20, 22):
            if lg > a.max_log2:
                continue
            n = 1 << lg
            ctx = Context(curve)
            ck = CommitterKey.generate(ctx, 1, n, ffi.AMSM_BASES_PRECOMPUTE)
            ffi.check(ctx._lib.amsm_bases_prebuild_twin(ctx._h, ck._h), "amsm_bases_prebuild_twin")
            t = ck.tables()
            v = [ctx.random_vector(5 + j, n, mont=False) for j in range(3)]
            VariableBaseMSM.multi_scalar_mul_batch(ck, v, mont=False)
            ws = ctx.memory()["workspace_bytes"]
            plain = n * (64 if curve == ffi.AMSM_PALLAS else 96)
            print(f"| {name} | 2^{lg} | {ck.window_bits} / {t['levels'] if 'levels' in t else '-'} | {mib(t['window_table'])} | {mib(t['direct_sum_table'])} | "
                  f"{mib(ck.memory()['twin'])} | {mib(plain)} | {mib(ws)} |", flush=True)
            for x in v:
                x.free()
            ck.free()
            ctx.close()


if __name__ == "__main__":
    main()

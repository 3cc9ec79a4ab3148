#!/usr/bin/env python3
"""Randomised parity of the AlignToReference tool (file to file) against the oracle's restatement.
Usage: python3 tools/fuzz_a2r.py [iterations] [seed]"""
import importlib.util, os, pathlib, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import calitas_amd as C
import oracle_lib as O

spec = importlib.util.spec_from_file_location("tk", os.path.join(ROOT, "tests", "test_gpu_kats.py"))
tk = importlib.util.module_from_spec(spec); spec.loader.exec_module(tk)


def run(iters, seed):
    rng = np.random.default_rng(seed)
    bad = 0
    t0 = time.time()
    ctx = C.Context(0)
    for it in range(iters):
        d = pathlib.Path(tempfile.mkdtemp(prefix="a2r"))
        fa, inp = tk._a2r_inputs(d, int(rng.integers(0, 1 << 30)))
        kw, okw = {}, {}
        if rng.random() < 0.6:
            dd, pp, oo, gg = int(rng.integers(0, 7)), int(rng.integers(0, 3)), int(rng.integers(0, 20)), int(rng.integers(0, 4))
            kw = dict(max_guide_diffs=dd, max_pam_mismatches=pp, max_overlap=oo, max_gaps_between_guide_and_pam=gg)
            okw = dict(limits=(dd, pp, oo), g=gg)
            if rng.random() < 0.4:
                D = int(rng.integers(0, dd + gg + pp + 1)); kw["max_total_diffs"] = D; okw["D"] = D
        else:
            gg = int(rng.integers(0, 4)); kw = dict(max_gaps_between_guide_and_pam=gg); okw = dict(g=gg)
        if rng.random() < 0.5:
            w = int(rng.choice([40, 60, 120, 300])); kw["window_size"] = w; okw["window_size"] = w
        try:
            header, want = O.align_to_reference(fa, inp, **okw)
        except RuntimeError as e:
            want = str(e)
        try:
            text = C.align_to_reference(inp, fa, None, version="unknown", time_stamp="n/a", **kw)
            lines = text.splitlines()
            got = [dict(zip(lines[0].split("\t"), ln.split("\t"))) for ln in lines[1:]]
        except Exception as e:
            got = str(e)
        if isinstance(want, str) or isinstance(got, str):
            if isinstance(want, str) != isinstance(got, str):
                bad += 1; print("ERROR MISMATCH iter %d %s: oracle %r product %r" % (it, kw, str(want)[:80], str(got)[:80]), flush=True)
            continue
        if got != want:
            bad += 1
            print("MISMATCH iter %d %s: %d vs %d rows" % (it, kw, len(got), len(want)), flush=True)
    ctx.close()
    print("fuzz_a2r: %d iterations, %d mismatches, %.1f s" % (iters, bad, time.time() - t0))
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 20, int(sys.argv[2]) if len(sys.argv) > 2 else 1) else 0)

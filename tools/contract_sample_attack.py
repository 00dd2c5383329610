#!/usr/bin/env python3
"""Directed attack on the contract-mode samplers (--ndf beckmann | ggx): candidates from several input families hill-climb
over the bit patterns of (u1, u2, o) to maximise the difference between the fp32 fast path and the bit-exact per-sample code,
in units of the contract (1e-5 max(1, |o|)).  The fast path is inside the contract as long as that score stays below 1 for
every sample it keeps; every evaluated kept sample outside it is counted (must be 0).
    PYTHONPATH=. python tools/contract_sample_attack.py [--m 262144] [--iters 512] [--rounds 2] > profiles/r04/contract_sample_attack.txt"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def families(m, dev, djb, synth, torch):
    R = lambda seed: djb.gen_uniforms(m, seed)
    base = djb.gen_directions(m, 41)
    yield "bench", R(42), R(43), base.clone()
    g = base.clone(); g[2] = 0.02 + 0.05 * g[2]; g /= g.norm(dim=0, keepdim=True)
    yield "grazing view", R(44), R(45), g
    nn = base.clone(); nn[0] *= 0.01; nn[1] *= 0.01; nn[2] = 1.0; nn /= nn.norm(dim=0, keepdim=True)
    yield "near-normal view", R(46), R(47), nn
    yield "u1 near 0", R(48) * 1e-3, R(49), base.clone()
    yield "u1 near 1", 1.0 - R(50) * 1e-3, R(51), base.clone()
    yield "u2 in the tails", torch.where(R(52) < 0.5, R(53) * 4e-3, 1.0 - R(53) * 4e-3), R(54), base.clone()
    yield "un-normalised view", R(55), R(56), base * (0.25 + 3.0 * R(57))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=1 << 18); ap.add_argument("--iters", type=int, default=512); ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--ndf", default="beckmann", choices=["beckmann", "ggx"])
    a = ap.parse_args()
    import torch
    from dj_brdf_amd import djb, synth
    ctx = djb.default_context(0); dev = f"cuda:{ctx.device}"
    b = getattr(djb, a.ndf)(ctx=ctx); P = djb.microfacet.params
    lobes = [("elliptic(0.2,0.5,0.7)", P.elliptic(0.2, 0.5, 0.7)), ("isotropic(0.05)", P.isotropic(0.05)), ("isotropic(0.02)", P.isotropic(0.02)), ("isotropic(1.0)", P.isotropic(1.0)),
             ("pdfparams(0.4,0.25,0.6,0.1,-0.2)", P.pdfparams(0.4, 0.25, 0.6, 0.1, -0.2))]
    tot_eval = tot_out = 0; worst = 0.0
    print(f"# contract_sample_attack ({a.ndf}): {a.m} candidates per family and lobe, {a.iters} moves x {a.rounds} rounds; score = component difference / (1e-5 max(1, |o|))")
    for lname, p in lobes:
        for name, u1, u2, o in families(a.m, dev, djb, synth, torch):
            u1, u2, o = u1.contiguous().float(), u2.contiguous().float(), o.contiguous().float()
            start, _ = djb.contract_sample_attack(b, u1.clone(), u2.clone(), o.clone(), p, iters=0, ctx=ctx)
            best = None
            for r in range(a.rounds):
                best, c = djb.contract_sample_attack(b, u1, u2, o, p, iters=a.iters, seed=23 + r, ctx=ctx)
                tot_eval += c["evaluations"]; tot_out += c["outside"]
                if r + 1 < a.rounds:
                    order = torch.argsort(best, descending=True)
                    top, bot = order[: a.m // 2], order[a.m // 2:]
                    u1[bot] = u1[top[: bot.numel()]]; u2[bot] = u2[top[: bot.numel()]]; o[:, bot] = o[:, top[: bot.numel()]]
            mx = float(best.max()); k = int(best.argmax()); worst = max(worst, mx)
            print(f"{lname:34s} {name:20s} start max {float(start.max()):.3f} -> attacked {mx:.3f}   kept outside the contract {c['outside']}   "
                  f"worst (u1, u2, o) = ({float(u1[k]):.9g}, {float(u2[k]):.9g}, {[float(x) for x in o[:, k]]})")
    print(f"# total evaluations {tot_eval:.3e}, kept samples outside the contract {tot_out}, worst score {worst:.3f} ({'OK: < 1' if worst < 1.0 and tot_out == 0 else 'ATTENTION'})")
    return 0 if worst < 1.0 and tot_out == 0 else 1


if __name__ == "__main__":
    sys.exit(main())

"""Worker for tests/test_host.py::test_two_rank_gloo_plan_exchange...: one CPU process per rank, gloo backend."""
import sys

import numpy as np
import torch.distributed as dist

import oracle as orc
from petsc_amd import dist as pdist

rank, world = int(sys.argv[1]), int(sys.argv[2])
kind = sys.argv[3] if len(sys.argv) > 3 else "27pt"
n = int(sys.argv[4]) if len(sys.argv) > 4 else 7
dist.init_process_group("gloo", rank=rank, world_size=world)
N = n ** 3
ranges = pdist.split_ownership(N, world)
rs, re = int(ranges[rank]), int(ranges[rank + 1])
ai, aj, aa = orc.stencil(kind, n, rs, re)
plan = pdist.build_plan(ai, aj, aa, ranges, rank, dist=dist)
# send / receive symmetry over all ranks: what rank a sends to b is exactly (ids and order) what b expects from a
allp = [None] * world
dist.all_gather_object(allp, {"rs": rs, "send_ranks": plan["send_ranks"], "send_off": plan["send_off"], "send_idx": plan["send_idx"],
                              "recv_ranks": plan["recv_ranks"], "recv_off": plan["recv_off"], "garray": plan["garray"]})
for a in range(world):
    pa = allp[a]
    for k, b in enumerate(pa["send_ranks"]):
        pb = allp[int(b)]
        kk = list(pb["recv_ranks"]).index(a)  # b must list a as a source
        sent_global = pa["rs"] + pa["send_idx"][pa["send_off"][k]:pa["send_off"][k + 1]]
        assert np.array_equal(sent_global, pb["garray"][pb["recv_off"][kk]:pb["recv_off"][kk + 1]]), (a, int(b))
    for kk, a2 in enumerate(pa["recv_ranks"]):
        assert a in list(allp[int(a2)]["send_ranks"]), (a, int(a2))
    assert len(set(map(int, pa["recv_ranks"]))) == len(pa["recv_ranks"]) and a not in pa["recv_ranks"] and a not in pa["send_ranks"]
# global x known everywhere; emulate the exchange: every rank packs what it was asked for, all ranks gather the packs
xg = 1.0 + (np.arange(N) % 17) / 17.0
xl = xg[rs:re]
packs = [None] * world
mine = {int(dst): xl[plan["send_idx"][plan["send_off"][k]:plan["send_off"][k + 1]]] for k, dst in enumerate(plan["send_ranks"])}
dist.all_gather_object(packs, mine)
lvec = np.zeros(plan["nghost"])
for k, src in enumerate(plan["recv_ranks"]):
    lvec[plan["recv_off"][k]:plan["recv_off"][k + 1]] = packs[int(src)][rank]
assert np.array_equal(lvec, xg[plan["garray"]])  # lvec[k] <-> garray[k] (mmaij.c:108-117)
# y = A_d x_l, then y += B_o lvec on the compressed rows (mpiaij.c:1056-1059), oracle arithmetic
m = plan["m"]
y = np.zeros(m)
L = orc.lib()
L.orc_MatMult_SeqAIJ(m, orc.P(plan["Ai"]), orc.P(plan["Aj"]), orc.P(plan["Aa"]), orc.P(np.ascontiguousarray(xl)), orc.P(y))
Bi_full = np.zeros(m + 1, np.int32)
cnt = np.zeros(m, np.int32)
cnt[plan["ridx"]] = np.diff(plan["Bi"])
Bi_full[1:] = np.cumsum(cnt)
z = np.zeros(m)
L.orc_MatMultAdd_SeqAIJ(m, orc.P(Bi_full), orc.P(plan["Bj"]), orc.P(plan["Ba"]), orc.P(lvec), orc.P(y), orc.P(z))
yref = orc.matmult(ai, aj, aa, xg)  # global columns on the slab = sequential product restricted to the slab
assert np.allclose(z, yref, rtol=1e-14, atol=1e-15), np.abs(z - yref).max()
dist.barrier()
dist.destroy_process_group()
print("PLAN_OK", rank)

"""LightGCN embedding propagation with the reference's interface (reference lightGCN.py:129-203).

`get_A_tilda` builds the symmetric-normalised bipartite adjacency D^-1/2 A D^-1/2 once on the host
(float32, as the reference), keeps it on the GPU as CSR and builds the static schedule of the SpMM
(spmm_bundle_plan: equal-cost runs of row bundles and hub-row pieces, one per resident wave; spmm_plan, the
first-generation virtual-row plan, for widths the bundled kernel does not take); `propagate_through_layers`
runs the n_layers SpMMs in the HIP kernel gdmcf_spmm_bundled_f32 (one launch per layer) with the layer mean
fused into the last layer's epilogue.
Only the forward propagation is on the hot path; BPR training is a
"next" row (8f3), so E0 gradients are not produced here.
"""
import ctypes

import numpy as np
import scipy.sparse as sp
import torch
import torch.nn as nn

from . import _lib

import os

SPMM_CHUNK = int(os.environ.get("GDMCF_SPMM_CHUNK", "256"))  # max nonzeros per virtual row


def normalized_bipartite_csr(users, items, n_users, n_items):
    """CSR (indptr int64, indices int32, data float32) of A~ = D^-1/2 [[0,R],[R^T,0]] D^-1/2,
    d = (rowsum + 1e-9)^-1/2 in float32, duplicates collapsed (reference :146-164)."""
    users = np.asarray(users, dtype=np.int64)
    items = np.asarray(items, dtype=np.int64)
    R = sp.coo_matrix((np.ones(len(users), np.float32), (users, items)), shape=(n_users, n_items)).tocsr()
    R.data[:] = 1.0
    A = sp.bmat([[None, R], [R.T.tocsr(), None]], format="csr", dtype=np.float32)
    A.sort_indices()
    N = n_users + n_items
    rowsum = np.asarray(A.sum(1), dtype=np.float32).flatten()
    d = np.power(rowsum + np.float32(1e-9), np.float32(-0.5)).astype(np.float32)
    d[np.isinf(d)] = 0.0
    rows = np.repeat(np.arange(N), np.diff(A.indptr))
    data = ((d[rows] * A.data).astype(np.float32) * d[A.indices]).astype(np.float32)
    return A.indptr.astype(np.int64), A.indices.astype(np.int32), data


SPMM_SHORT = int(os.environ.get("GDMCF_SPMM_SHORT", "48"))  # rows up to this many nonzeros run four-to-a-wave


def spmm_plan(indptr, chunk=SPMM_CHUNK, short=SPMM_SHORT, d=64):
    """Virtual-row plan for gdmcf_spmm_csr_f32 (see include/gdmcf_hip.h).  Rows with <= `short` nonzeros come
    first, whole (they run 64/(d/4) to a wave); every other row is cut into pieces of <= chunk nonzeros; rows cut
    into more than one piece get partial slots + an entry in lrow/lptr."""
    indptr = np.asarray(indptr, dtype=np.int64)
    deg = np.diff(indptr)
    lpr = d // 4
    if d % 4 != 0 or lpr not in (2, 4, 8, 16, 32, 64):
        short = -1  # the short-row kernel needs power-of-two lane groups
    is_short = deg <= short
    srows = np.nonzero(is_short)[0]
    lrows_all = np.nonzero(~is_short)[0]
    ldeg = deg[lrows_all]
    pieces = np.maximum(1, -(-ldeg // chunk)).astype(np.int64)
    n_rest = int(pieces.sum())
    rrow = np.repeat(lrows_all, pieces)
    first = np.concatenate([[0], np.cumsum(pieces)[:-1]]) if len(pieces) else np.zeros(0, np.int64)
    k = np.arange(n_rest, dtype=np.int64) - np.repeat(first, pieces)  # piece index inside its row
    rbeg = indptr[rrow] + k * chunk
    rend = np.minimum(rbeg + chunk, indptr[rrow + 1])
    cut = np.repeat(pieces > 1, pieces)
    rslot = np.full(n_rest, -1, dtype=np.int32)
    rslot[cut] = np.arange(int(cut.sum()), dtype=np.int32)
    lrow = lrows_all[pieces > 1].astype(np.int32)
    lptr = np.concatenate([[0], np.cumsum(pieces[pieces > 1])]).astype(np.int32)
    return dict(vbeg=np.concatenate([indptr[srows], rbeg]).astype(np.int64),
                vend=np.concatenate([indptr[srows + 1], rend]).astype(np.int64),
                vrow=np.concatenate([srows, rrow]).astype(np.int32),
                vslot=np.concatenate([np.full(len(srows), -1, np.int32), rslot]).astype(np.int32),
                lrow=lrow, lptr=lptr, n_slots=int(cut.sum()), n_short=int(len(srows)))


SPMM_SMAX = int(os.environ.get("GDMCF_SPMM_SMAX", "64"))  # rows up to this many nonzeros are bundled G to a wave-step
SPMM_PIECE = int(os.environ.get("GDMCF_SPMM_PIECE", "128"))  # longer rows are cut into pieces of at most this many
SPMM_WAVES = int(os.environ.get("GDMCF_SPMM_WAVES", "4096"))  # 256 CUs x 16 resident waves, all started at once
# fixed cost of a bundled row / of a piece, in nonzeros.  Calibrated on the per-wave timeline of the Yelp-shape graph
# (tools/spmm_waves.py: least squares over 4096 waves): 0.56 us per piece group of 16 nonzeros, 0.67 / 0.76 us per warm /
# cold bundle group, 0.42 us per piece, 1.0-1.1 us per bundle of four rows
SPMM_COST_ROW = int(os.environ.get("GDMCF_SPMM_COST_ROW", "8"))
SPMM_COST_PIECE = int(os.environ.get("GDMCF_SPMM_COST_PIECE", "12"))
SPMM_HOT_MB = float(os.environ.get("GDMCF_SPMM_HOT_MB", "2.0"))  # part of an XCD's L2 the bundled rows' hot columns can keep
SPMM_SLICE_MB = float(os.environ.get("GDMCF_SPMM_SLICE_MB", "3.5"))  # largest per-XCD slice of the table worth cutting rows for
SPMM_NT = int(os.environ.get("GDMCF_SPMM_NT", "0"))  # streaming loads for bundles that gather mostly cold rows
SPMM_MISS_W = float(os.environ.get("GDMCF_SPMM_MISS_W", "1.15"))  # cost of a gather of a rarely gathered row, in warm ones


def spmm_bundle_plan(indptr, indices, d=64, n_waves=None, s_max=None, piece=None, classes=8, n_cols=None, split_at=None):
    """Static schedule for gdmcf_spmm_bundled_f32 (include/gdmcf_hip.h; kernel csrc/spmm_bundle.hip).

    Work units: every row of at most s_max nonzeros ("short"), and the pieces (<= piece nonzeros, equal parts) of every
    longer row.  Class c (of `classes` = 8) is served by the blocks b with b % 8 == c, i.e. by ONE XCD and its 4 MiB L2:
    (1) the column space is cut into 8 ranges holding equal numbers of the long rows' nonzeros; a long row is cut where
        it crosses a range boundary (CSR rows are sorted) and class = range: the pieces of a class gather 1/8 of the
        table (2-3 MB at the Yelp shape: L2 resident), at the price of one 256-byte partial per piece;
    (2) the short rows are sorted by length (longest first) and dealt to the classes bundle by bundle (G = 64/(d/4) rows
        of the same length share a wave-step, so the lane groups of a wave finish together); every class gets the same
        mix, so the gathers that miss L2 load the eight fabric links equally; rows whose columns are mostly among the
        most gathered ones ("warm") and the others ("cold") are bundled apart;
    (3) every wave of a class gets an equal-cost share of the class's pieces, of its warm and of its cold bundles, and
        runs them in this order: the XCD's waves are in the same phase at the same time.
    Cost = nonzeros (those to a column outside the `SPMM_HOT_MB` most gathered megabytes of the table count SPMM_MISS_W
    times: they are served by the Infinity Cache at a third of the L2 rate) + a fixed overhead per row / piece.
    Rows cut into several pieces get consecutive partial slots (crow / cptr), added up in slot order afterwards.
    `split_at` (round 4 experiment, GDMCF_SPMM_SPLIT=1): rows below it (the users of the bipartite graph) and rows from it on
    (the items) are bundled and PHASED apart -- user rows gather only the item half of the table and vice versa, so that all
    XCDs gather from one half at a time; "most gathered" is then judged per half."""
    s_max = SPMM_SMAX if s_max is None else s_max
    piece = SPMM_PIECE if piece is None else piece
    indptr = np.asarray(indptr, dtype=np.int64)
    indices = np.asarray(indices)
    deg = np.diff(indptr)
    lpr = d // 4
    if d % 4 != 0 or lpr not in (2, 4, 8, 16, 32, 64):
        raise ValueError("spmm_bundle_plan: d must be one of 8, 16, 32, 64, 128, 256")
    G = 64 // lpr
    n_cols = int(indices.max()) + 1 if (n_cols is None and len(indices)) else int(n_cols or 1)
    # ---- which gathers can hit: the most gathered rows of the table, as many as fit beside the pieces' slice ----
    ccount = np.bincount(indices, minlength=n_cols)
    n_hot = int(SPMM_HOT_MB * 1e6 // (d * 4))
    if split_at is not None and 0 < split_at < n_cols:
        # per half of the table: the n_hot most gathered rows of the half that is being gathered from
        cold_col = np.zeros(n_cols, bool)
        for lo, hi in ((0, int(split_at)), (int(split_at), n_cols)):
            cc = ccount[lo:hi]
            if n_hot < hi - lo:
                th = np.partition(cc, hi - lo - n_hot)[hi - lo - n_hot]
                cold_col[lo:hi] = cc < max(th, 1)
        cold = cold_col[indices]
    elif n_hot < n_cols:
        thresh = np.partition(ccount, n_cols - n_hot)[n_cols - n_hot]
        cold = (ccount < max(thresh, 1))[indices]
    else:
        cold = np.zeros(len(indices), bool)
    csum = np.concatenate([[0], np.cumsum(cold, dtype=np.int64)])
    ncold = csum[indptr[1:]] - csum[indptr[:-1]]  # per row
    wdeg = deg + (SPMM_MISS_W - 1.0) * ncold  # weighted nonzeros
    srows = np.nonzero(deg <= s_max)[0]
    lrows = np.nonzero(deg > s_max)[0]
    # ---- (1) pieces of the long rows: cut where the row crosses one of `classes` column ranges (each holding the same
    # number of the long rows' nonzeros), then into equal parts of <= piece nonzeros; class = the range ----
    lmask = np.zeros(len(deg), bool)
    lmask[lrows] = True
    row_of = np.repeat(np.arange(len(deg)), deg)
    lidx = np.nonzero(lmask[row_of])[0]  # CSR positions of the long rows' nonzeros, in CSR order
    if len(lidx):
        lr, lc = row_of[lidx], indices[lidx]
        bounds = np.quantile(lc, np.arange(1, classes) / classes, method="lower").astype(np.int64)
        rng = np.searchsorted(bounds, lc, side="left")  # 0 .. classes-1, non-decreasing along a (sorted) row
        start = np.concatenate([[True], (lr[1:] != lr[:-1]) | (rng[1:] != rng[:-1])])
        rs = np.nonzero(start)[0]  # runs: (row, range)
        rlen = np.diff(np.concatenate([rs, [len(lidx)]]))
        rbeg, rrow, rrng = lidx[rs], lr[rs], rng[rs]
        npc = -(-rlen // piece)  # parts per run
        psz = -(-rlen // npc)
        n_p = int(npc.sum())
        first = np.cumsum(npc) - npc
        k = np.arange(n_p, dtype=np.int64) - np.repeat(first, npc)
        pbeg = np.repeat(rbeg, npc) + k * np.repeat(psz, npc)
        plen = np.minimum(np.repeat(psz, npc), np.repeat(rbeg + rlen, npc) - pbeg).astype(np.int64)
        prow = np.repeat(rrow, npc)
        pcls = np.repeat(rrng, npc)
        ppr = np.bincount(prow, minlength=len(deg))  # pieces per row
        cut = ppr[prow] > 1
        pslot = np.full(n_p, -1, dtype=np.int64)
        pslot[cut] = np.arange(int(cut.sum()))  # CSR order: the slots of a row are consecutive
        crow = np.nonzero(ppr > 1)[0].astype(np.int32)
        cptr = np.concatenate([[0], np.cumsum(ppr[crow])]).astype(np.int32)
    else:
        n_p = 0
        pbeg = plen = prow = pcls = pslot = np.zeros(0, np.int64)
        cut = np.zeros(0, bool)
        crow, cptr = np.zeros(0, np.int32), np.zeros(1, np.int32)
    ptot = float((plen + float(SPMM_COST_PIECE)).sum())
    pord = np.lexsort((pbeg, pcls))  # class-major, CSR order inside
    P = dict(beg=pbeg[pord], len=plen[pord], row=prow[pord], slot=pslot[pord], cls=pcls[pord])
    # (2) short rows: "warm" rows (most of their columns are among the most gathered ones) and "cold" rows apart, each
    # set longest first and dealt to the classes G rows (= one bundle) at a time
    cold_row = (ncold[srows] * 2 > deg[srows]).astype(np.int64)
    NG = 2  # bundle groups per class = phases after the pieces: (warm, cold), or (users warm, users cold, items warm, items cold)
    if split_at is not None:
        NG = 4
        cold_row = cold_row + 2 * (srows >= int(split_at)).astype(np.int64)
    sord = np.lexsort((srows, -deg[srows], cold_row))  # group by group, longest first
    s_sorted, s_temp = srows[sord], cold_row[sord]
    n_key = np.bincount(s_temp, minlength=NG)
    key_first = np.cumsum(n_key) - n_key
    pos_in_temp = np.arange(len(s_sorted)) - key_first[s_temp]
    # Dealing: class c takes the share of the bundles that fills it up to the same total cost as the others (the pieces
    # of a class that straddles two kinds of rows are shorter and more numerous, so that class gets fewer bundles) --
    # stride scheduling: class c picks at times (j + 1/2) / share_c, bundle k goes to whoever picks k-th.
    pcost_cls = np.bincount(P["cls"], weights=P["len"] + float(SPMM_COST_PIECE), minlength=classes) if n_p else np.zeros(classes)
    rough_total = pcost_cls.sum() + float((wdeg[srows] + SPMM_COST_ROW).sum())
    room = np.maximum(rough_total / classes - pcost_cls, 0.02 * rough_total / classes)
    share = room / room.sum()

    def deal(n_bundles):
        if n_bundles == 0:
            return np.zeros(0, np.int64)
        quota = np.maximum(np.floor(share * n_bundles).astype(np.int64), 0)
        quota[np.argsort(-(share * n_bundles - quota))[: n_bundles - quota.sum()]] += 1
        times = np.concatenate([(np.arange(q_) + 0.5) / max(q_, 1) for q_ in quota])
        who = np.repeat(np.arange(classes), quota)
        return who[np.argsort(times, kind="stable")]

    cls_of_bundle = [deal(-(-int(n_key[k_]) // G)) for k_ in range(NG)]
    s_cls = np.zeros(len(s_sorted), dtype=np.int64)
    for k_ in range(NG):
        sel = s_temp == k_
        if n_key[k_]:
            s_cls[sel] = cls_of_bundle[k_][np.minimum(pos_in_temp[sel] // G, len(cls_of_bundle[k_]) - 1)]
    grp = s_cls * NG + s_temp  # (class, group) pairs, bundles never mix them
    gord = np.argsort(grp, kind="stable")
    s_row, s_grp = s_sorted[gord], grp[gord]
    n_grp_rows = np.bincount(s_grp, minlength=NG * classes)
    nb_grp = -(-n_grp_rows // G)  # bundles per group (last one padded)
    n_b = int(nb_grp.sum())
    gb_first = np.cumsum(nb_grp) - nb_grp
    gr_first = np.cumsum(n_grp_rows) - n_grp_rows
    ent0 = gb_first[s_grp] * G + (np.arange(len(s_row)) - gr_first[s_grp])  # entry index, group-major bundle order
    b_grp = np.repeat(np.arange(NG * classes), nb_grp)
    slen0 = np.zeros(n_b * G, dtype=np.int64)
    swd0 = np.zeros(n_b * G, dtype=np.float64)
    slen0[ent0], swd0[ent0] = deg[s_row], wdeg[s_row]
    bcost0 = (swd0.reshape(n_b, G).max(axis=1) + float(SPMM_COST_ROW)) * G if n_b else np.zeros(0)
    total = ptot + float(bcost0.sum())
    if n_waves is None:
        n_waves = int(min(SPMM_WAVES, max(32, -(-int(total) // 4096) * 32)))  # >= ~128 cost units per wave
    wpc = n_waves // classes
    assert n_waves % (4 * classes) == 0

    # (3) every wave of a class takes a share of EACH phase -- the class's pieces, its warm bundles, its cold bundles --
    # and runs them in that order, so that all waves of the XCD gather from the pieces' slice of the table first (L2
    # resident as long as nothing else streams through), then from the much-gathered rows, and last do the gathers that
    # miss anyway.  (Pieces, warm and cold bundles run side by side evicted each other: 46 % L2 hits.)  Units are
    # indivisible and a wave's share of one phase is only a dozen gather groups (one bundle of 64-nonzero rows is 16),
    # so the shares are dealt greedily on the RUNNING total: phase by phase, largest unit first, to the wave that holds
    # the least so far (longest-processing-time rule) -- the waves' totals stay within one small unit of each other
    # after every phase, which also keeps the phases aligned in time.
    import heapq
    pcost_sorted = P["len"] + float(SPMM_COST_PIECE)
    p_wave = np.zeros(n_p, dtype=np.int64)
    b_wave = np.zeros(n_b, dtype=np.int64)
    for c in range(classes):
        heap = [(0.0, j) for j in range(wpc)]
        for phase in range(1 + NG):
            if phase == 0:
                idx = np.nonzero(P["cls"] == c)[0]
                cost = pcost_sorted[idx]
            else:
                idx = np.nonzero(b_grp == c * NG + (phase - 1))[0]
                cost = bcost0[idx]
            order_ = np.argsort(-cost, kind="stable")
            dest = np.empty(len(idx), dtype=np.int64)
            for k_, cst in zip(order_.tolist(), cost[order_].tolist()):
                tot_, j = heapq.heappop(heap)
                dest[k_] = j
                heapq.heappush(heap, (tot_ + cst, j))
            if phase == 0:
                p_wave[idx] = c * wpc + dest
            else:
                b_wave[idx] = c * wpc + dest
    pw = np.argsort(p_wave, kind="stable")  # wave-major, CSR order inside
    P = {k_: v_[pw] for k_, v_ in P.items()}
    p_wave = p_wave[pw]
    bw = np.lexsort((np.arange(n_b), b_grp % NG, b_wave))  # wave-major, phase by phase, longest first
    new_of_old = np.empty(n_b, dtype=np.int64)
    new_of_old[bw] = np.arange(n_b)
    ent = new_of_old[ent0 // G] * G + ent0 % G
    sbeg = np.zeros(n_b * G, dtype=np.int64)
    slen = np.zeros(n_b * G, dtype=np.int32)
    srow = np.full(n_b * G, -1, dtype=np.int32)
    sbeg[ent], slen[ent], srow[ent] = indptr[s_row], deg[s_row], s_row
    smax = slen.reshape(n_b, G).max(axis=1).astype(np.int32) if n_b else np.zeros(0, np.int32)
    if n_b and SPMM_NT:  # bit 30: the bundle's rows gather mostly rarely gathered rows -> streaming (nt) loads
        smax = smax | ((b_grp[bw] & 1).astype(np.int32) << 30)
    wdesc = np.zeros((n_waves, 4), dtype=np.int32)
    pcnt = np.bincount(p_wave, minlength=n_waves)
    bcnt = np.bincount(b_wave, minlength=n_waves)
    wdesc[:, 1] = np.cumsum(pcnt)
    wdesc[:, 0] = wdesc[:, 1] - pcnt
    wdesc[:, 3] = np.cumsum(bcnt)
    wdesc[:, 2] = wdesc[:, 3] - bcnt
    return dict(wdesc=wdesc.reshape(-1), n_waves=int(n_waves), lbeg=P["beg"].astype(np.int64), llen=P["len"].astype(np.int32),
                lrow=P["row"].astype(np.int32), lslot=P["slot"].astype(np.int32), n_pieces=n_p, sbeg=sbeg, slen=slen, srow=srow,
                smax=smax, n_bundles=n_b, crow=crow, cptr=cptr, n_slots=int(cut.sum()), G=G)


def spmm_stream_pack(plan, indptr, indices, vals, d=64):
    """Packs the schedule of spmm_bundle_plan into what the kernel streams (csrc/spmm_bundle.hip, gdmcf_spmm_stream_f32):
    the nonzeros themselves, re-ordered so that every wave reads ONE contiguous run of (col, val) pairs in exactly the order
    it gathers them, and one small descriptor per unit.  A wave that has to chase row pointers (descriptor -> col/val of
    that row -> gathers) pays an HBM latency per row; with a contiguous run it touches its lines once when it starts and
    afterwards only ever waits for L2.

    Layout: a unit (a piece, or a bundle of G rows) is a sequence of steps, a step = G entries, entry (k, g) = what lane
    group g gathers in step k; pieces put nonzero j at (j // G, j % G), bundles put the k-th nonzero of row g at (k, g).
    Units are padded to a multiple of UN = min(4, d/4) steps and a wave's run to a multiple of 64 entries with entries of
    weight 0 that point at a column the row really has (empty rows: flagged in the descriptor, the sum is discarded).
      cw    int32 [n_entries, 2]   (col, float bits of val)
      ud    int32 [n_units, DW]    DW = 1 + max(G, 2): [steps/UN | kind << 31, piece: row, slot | bundle: row_g (-1 =
                                   padding, bit 30 = empty row)]
      wdesc int32 [n_waves, 4]     first batch (64 entries) of the wave's run, its batches, first / last+1 unit"""
    indptr = np.asarray(indptr, dtype=np.int64)
    indices = np.asarray(indices)
    vals = np.asarray(vals, dtype=np.float32)
    nnz = len(indices)
    lpr = d // 4
    G = 64 // lpr
    UN = min(4, lpr)
    DW = 1 + max(G, 2)
    n_p, n_b, n_w = plan["n_pieces"], plan["n_bundles"], plan["n_waves"]
    wd = plan["wdesc"].reshape(n_w, 4).astype(np.int64)
    # units in wave order: the pieces of a wave, then its bundles
    p_wave = np.repeat(np.arange(n_w), wd[:, 1] - wd[:, 0])
    b_wave = np.repeat(np.arange(n_w), wd[:, 3] - wd[:, 2])
    n_u = n_p + n_b
    pu = np.arange(n_p) + wd[p_wave, 2]  # unit id of a piece: the bundles of the earlier waves come before it
    bu = np.arange(n_b) + wd[b_wave, 1]  # unit id of a bundle: the pieces up to and including its own wave's
    # ---- steps per unit ----
    llen = plan["llen"].astype(np.int64)
    smax = (plan["smax"] & 0x3FFFFFFF).astype(np.int64)
    usteps = np.zeros(n_u, dtype=np.int64)
    usteps[pu] = -(-(-(-llen // G)) // UN) * UN
    usteps[bu] = -(-smax // UN) * UN
    uent = usteps * G
    # ---- waves: unit ranges, stream offsets (batch aligned) ----
    u0 = wd[:, 0] + wd[:, 2]
    u1 = wd[:, 1] + wd[:, 3]
    ucum = np.concatenate([[0], np.cumsum(uent)])
    wtot = ucum[u1] - ucum[u0]
    wbat = -(-wtot // 64)
    # a wave whose units are all bundles of EMPTY rows has no entries of its own; the kernel still pre-loads the first
    # batch of its run, so every wave with units owns at least one (weight-0, column-0) batch
    wbat = np.where((u1 > u0) & (wbat == 0), 1, wbat)
    wbase = (np.cumsum(wbat) - wbat) * 64
    n_ent = int(wbat.sum()) * 64
    wave_of_unit = np.repeat(np.arange(n_w), u1 - u0)  # units are in wave order
    upos = wbase[wave_of_unit] + (ucum[:-1] - ucum[u0][wave_of_unit])
    wdesc = np.stack([wbase // 64, wbat, u0, u1], axis=1).astype(np.int32)
    # ---- entries ----
    c_out = np.zeros(max(n_ent, 1), dtype=np.int32)
    w_out = np.zeros(max(n_ent, 1), dtype=np.float32)
    if n_p:
        ent = uent[pu]
        first = np.cumsum(ent) - ent
        j = np.arange(int(ent.sum()), dtype=np.int64) - np.repeat(first, ent)
        ln = np.repeat(llen, ent)
        src = np.repeat(plan["lbeg"].astype(np.int64), ent) + np.minimum(j, ln - 1)
        pos = np.repeat(upos[pu], ent) + j
        c_out[pos] = indices[src]
        w_out[pos] = np.where(j < ln, vals[src], np.float32(0))
    if n_b:
        ent = uent[bu]
        first = np.cumsum(ent) - ent
        loc = np.arange(int(ent.sum()), dtype=np.int64) - np.repeat(first, ent)
        bidx = np.repeat(np.arange(n_b), ent)
        k, g = loc // G, loc % G
        e = bidx * G + g
        rl = plan["slen"].astype(np.int64)[e]
        src = np.clip(plan["sbeg"].astype(np.int64)[e] + np.minimum(k, np.maximum(rl - 1, 0)), 0, max(nnz - 1, 0))
        pos = np.repeat(upos[bu], ent) + loc
        c_out[pos] = indices[src] if nnz else 0
        w_out[pos] = np.where(k < rl, vals[src], np.float32(0)) if nnz else 0
    # wave tails (padding up to a whole batch): any column of the matrix, weight 0 -- already 0 / 0.0 (column 0 exists)
    cw = np.stack([c_out, w_out.view(np.int32)], axis=1)
    # ---- unit descriptors ----
    ud = np.full((max(n_u, 1), DW), -1, dtype=np.int32)
    if n_p:
        ud[pu, 0] = (usteps[pu] // UN).astype(np.int32) | np.int32(-2147483648)
        ud[pu, 1] = plan["lrow"]
        ud[pu, 2] = plan["lslot"]
    if n_b:
        ud[bu, 0] = (usteps[bu] // UN).astype(np.int32)
        rows = plan["srow"].reshape(n_b, G).astype(np.int32)
        empty = (plan["slen"].reshape(n_b, G) == 0) & (rows >= 0)
        ud[bu, 1:1 + G] = np.where(empty, rows | np.int32(1 << 30), rows)
    return dict(cw=cw.reshape(-1), ud=ud.reshape(-1), wdesc=wdesc.reshape(-1), n_waves=n_w, n_units=n_u, n_entries=n_ent,
                crow=plan["crow"], cptr=plan["cptr"], n_slots=plan["n_slots"], G=G, UN=UN, DW=DW)


class LightGCN(nn.Module):
    def __init__(self, data, n_users, n_items, n_layers, latent_dim, device="cuda", shard_rows=False, group=None):
        """data: DataFrame/dict with `user_id_idx` / `item_id_idx` columns (reference :147).

        shard_rows (new, multi-GPU; the reference is single-device): every rank of `group` keeps only a contiguous
        block of ceil(N/world) rows of the normalised adjacency.  A layer is then a local SpMM on the full embedding
        table of the previous layer plus an all-gather of the row blocks (N*d*4/world bytes per rank per layer over
        xGMI); the backward pass all-reduces the cotangent first -- propagation is linear, so sum_r A~^l g_r =
        A~^l sum_r g_r -- which makes E0.grad the SUM over ranks already (do not all-reduce it again)."""
        super().__init__()
        import torch.distributed as dist
        self._group = group
        self._world = dist.get_world_size(group) if (shard_rows and dist.is_initialized()) else 1
        self._rank = dist.get_rank(group) if self._world > 1 else 0
        self.data = data
        self.n_users, self.n_items = n_users, n_items
        self.n_layers, self.latent_dim = n_layers, latent_dim
        self._device = torch.device(device)
        self.init_embedding()
        self.norm_adj_csr = self.get_A_tilda()

    def init_embedding(self):
        self.E0 = nn.Embedding(self.n_users + self.n_items, self.latent_dim)
        nn.init.xavier_uniform_(self.E0.weight)
        self.E0.weight = nn.Parameter(self.E0.weight)

    def get_A_tilda(self):
        indptr, indices, vals = normalized_bipartite_csr(np.asarray(self.data["user_id_idx"]),
                                                         np.asarray(self.data["item_id_idx"]), self.n_users,
                                                         self.n_items)
        dev = self._device
        N = self.n_users + self.n_items
        rpr = -(-N // self._world)  # rows per rank
        r0 = min(self._rank * rpr, N)
        r1 = min(r0 + rpr, N)
        self._rows = (r0, r1, rpr)
        if self._world > 1:
            lo, hi = int(indptr[r0]), int(indptr[r1])
            indptr, indices, vals = indptr[r0:r1 + 1] - indptr[r0], indices[lo:hi], vals[lo:hi]
        lpr = self.latent_dim // 4
        fits = self.latent_dim % 4 == 0 and lpr in (2, 4, 8, 16, 32, 64) and indices.size > 0
        # Which kernel: the streamed schedule (third generation) pays when an eighth of the gathered table -- what one XCD
        # gathers through its pieces -- fits that XCD's 4 MiB L2 beside the streams (Yelp shape: 2.85 MB; 56 us per layer
        # against 61 us); on larger tables its length-sorted bundles scatter the 256-byte result rows over the whole
        # output and the row-ordered first-generation kernels are faster (stress shape: 1.29 ms against 1.37 ms).
        gen = os.environ.get("GDMCF_SPMM_GEN", "auto")
        if gen == "auto":
            gen = "3" if N * self.latent_dim * 4 / 8 <= SPMM_SLICE_MB * 1e6 else "1"
        self._bundled = fits and gen == "2"
        self._streamed = fits and gen == "3"
        if self._streamed:
            split = self.n_users if (os.environ.get("GDMCF_SPMM_SPLIT", "0") == "1" and self._world == 1) else None
            plan = spmm_stream_pack(spmm_bundle_plan(indptr, indices, d=self.latent_dim, n_cols=N, split_at=split), indptr, indices,
                                    vals, d=self.latent_dim)
        elif self._bundled:
            plan = spmm_bundle_plan(indptr, indices, d=self.latent_dim, n_cols=N)
        else:
            plan = spmm_plan(indptr, d=self.latent_dim)
        self._plan = {k: (torch.from_numpy(v).to(dev) if isinstance(v, np.ndarray) else v) for k, v in plan.items()}
        self._partial = torch.empty(max(plan["n_slots"], 1), self.latent_dim, dtype=torch.float32, device=dev)
        self.nnz = int(indices.size)
        return (torch.from_numpy(indptr).to(dev), torch.from_numpy(indices).to(dev), torch.from_numpy(vals).to(dev))

    def algorithmic_bytes(self):
        """Compulsory HBM bytes of one layer: CSR once, X once, Y once."""
        N, d = self.n_users + self.n_items, self.latent_dim
        rows = self._rows[1] - self._rows[0]  # this rank's block: CSR once, X once, its rows of Y once
        return self.nnz * 8.0 + (rows + 1) * 8.0 + (N + rows) * d * 4.0

    @torch.no_grad()
    def _propagate(self, X, return_layers=False):
        """mean_l (A~^l X), l = 0..n_layers, through gdmcf_spmm_csr_f32 (layer mean fused into the last SpMM)."""
        lib = _lib.load()
        _lib.require_gpu(X, "LightGCN embeddings")
        _, indices, vals = self.norm_adj_csr
        pl = self._plan
        N, d = X.shape
        st = _lib.stream_ptr()
        cur = X.contiguous()
        layers = [cur]
        nv, nl = (0, 0) if (self._bundled or self._streamed) else (pl["vrow"].numel(), pl["lrow"].numel())
        r0, r1, rpr = self._rows
        sharded = self._world > 1
        for layer in range(self.n_layers):
            last = (layer == self.n_layers - 1) and not return_layers
            out = torch.zeros(rpr, d, dtype=cur.dtype, device=cur.device) if sharded else torch.empty_like(cur)
            adds = layers if last else []
            # the fused layer mean reads this rank's rows of the earlier layers
            arr = (ctypes.c_void_p * max(len(adds), 1))(*[a.data_ptr() + r0 * a.stride(0) * 4 for a in adds])
            if r1 > r0 and self._streamed:
                opt = lambda t: t.data_ptr() if t.numel() else None
                _lib.check(lib.gdmcf_spmm_stream_f32(
                    pl["wdesc"].data_ptr(), pl["n_waves"], pl["cw"].data_ptr(), pl["n_entries"], pl["ud"].data_ptr(),
                    pl["n_units"], opt(pl["crow"]), pl["cptr"].data_ptr(), pl["crow"].numel(), r1 - r0, N, cur.data_ptr(),
                    cur.stride(0), d, out.data_ptr(), out.stride(0), self._partial.data_ptr(), arr, len(adds), cur.stride(0),
                    1.0 / (self.n_layers + 1) if last else 1.0, self.algorithmic_bytes(), st))
            elif r1 > r0 and self._bundled:
                opt = lambda t: t.data_ptr() if t.numel() else None
                _lib.check(lib.gdmcf_spmm_bundled_f32(
                    pl["wdesc"].data_ptr(), pl["n_waves"], opt(pl["lbeg"]), opt(pl["llen"]), opt(pl["lrow"]), opt(pl["lslot"]),
                    pl["n_pieces"], opt(pl["sbeg"]), opt(pl["slen"]), opt(pl["srow"]), opt(pl["smax"]), pl["n_bundles"],
                    opt(pl["crow"]), pl["cptr"].data_ptr(), pl["crow"].numel(), indices.data_ptr(), vals.data_ptr(),
                    indices.numel(), r1 - r0, N, cur.data_ptr(), cur.stride(0), d, out.data_ptr(), out.stride(0), self._partial.data_ptr(), arr,
                    len(adds), cur.stride(0), 1.0 / (self.n_layers + 1) if last else 1.0, self.algorithmic_bytes(), st))
            elif r1 > r0:
                _lib.check(lib.gdmcf_spmm_csr_f32(
                    pl["vbeg"].data_ptr(), pl["vend"].data_ptr(), pl["vrow"].data_ptr(), pl["vslot"].data_ptr(), nv,
                    pl["n_short"], _lib.ptr(pl["lrow"]) if nl else None,
                    _lib.ptr(pl["lptr"]) if nl else None, nl, indices.data_ptr(), vals.data_ptr(), r1 - r0,
                    cur.data_ptr(), cur.stride(0), d, out.data_ptr(), out.stride(0), self._partial.data_ptr(), arr,
                    len(adds), cur.stride(0), 1.0 / (self.n_layers + 1) if last else 1.0, self.algorithmic_bytes(), st))
            if sharded:
                from .parallel import all_gather_rows
                out = all_gather_rows(out, N, self._group)
            layers.append(out)
            cur = out
        if return_layers:
            mean = torch.stack(layers).sum(0) / (self.n_layers + 1)  # test/debug path only
            return mean, layers[1:]
        return (cur if self.n_layers > 0 else X), None

    def propagate_through_layers(self, return_layers=False):
        """(final_user, final_item, initial_user, initial_item) as reference lightGCN.py:180-194.  Differentiable
        w.r.t. E0 when autograd is on (the backward is the same propagation: A~ is symmetric)."""
        E0 = self.E0.weight
        if return_layers:
            mean, layers = self._propagate(E0.detach(), return_layers=True)
        elif torch.is_grad_enabled() and E0.requires_grad:
            mean, layers = _PropagateMean.apply(self, E0), None
        else:
            mean, layers = self._propagate(E0.detach())[0], None
        final_user, final_item = torch.split(mean, [self.n_users, self.n_items])
        init_user, init_item = torch.split(E0 if layers is None else E0.detach(), [self.n_users, self.n_items])
        if return_layers:
            return final_user, final_item, init_user, init_item, layers
        return final_user, final_item, init_user, init_item

    def forward(self, users, pos_items, neg_items):
        fu, fi, iu, ii = self.propagate_through_layers()
        return fu[users], fi[pos_items], fi[neg_items], iu[users], ii[pos_items], ii[neg_items]


class _PropagateMean(torch.autograd.Function):
    """mean over layers of A~^l E0.  d(mean)/d(E0) applied to a cotangent G is mean_l (A~^l)^T G = mean_l A~^l G
    because the normalised adjacency is symmetric (reference lightGCN.py:149-164): the backward pass is the
    forward kernel run on the gradient."""

    @staticmethod
    def forward(ctx, module, E0):
        ctx.module = module
        return module._propagate(E0.detach())[0]

    @staticmethod
    def backward(ctx, g):
        m = ctx.module
        g = g.contiguous()
        if m._world > 1:  # row-sharded: propagate the SUM of all ranks' cotangents (linearity), see LightGCN.__init__
            from .parallel import _all_reduce
            g = g.clone()
            _all_reduce(g, m._group)
        return None, m._propagate(g)[0]


def bpr_loss(users, users_emb, pos_emb, neg_emb, userEmb0, posEmb0, negEmb0):
    """(mf_loss, reg_loss) of reference lightGCN.py:207-219."""
    reg_loss = (1 / 2) * (userEmb0.norm().pow(2) + posEmb0.norm().pow(2) + negEmb0.norm().pow(2)) / float(len(users))
    pos_scores = torch.sum(torch.mul(users_emb, pos_emb), dim=1)
    neg_scores = torch.sum(torch.mul(users_emb, neg_emb), dim=1)
    loss = torch.mean(torch.nn.functional.softplus(neg_scores - pos_scores))
    return loss, reg_loss


def sample_bpr_batch(indptr, indices, n_users, n_items, batch_size, rng):
    """(users, pos_items, neg_items) as the reference sampler (lightGCN.py:221-251, minus its stray
    pdb.set_trace): sorted users drawn without replacement (with replacement if n_users < batch_size), one random
    interacted item and one random non-interacted item per user.  Vectorised numpy; `rng` = np.random.Generator."""
    if n_users < batch_size:
        users = np.sort(rng.integers(0, n_users, batch_size))
    else:
        users = np.sort(rng.choice(n_users, batch_size, replace=False))
    deg = indptr[users + 1] - indptr[users]
    if np.any(deg == 0):
        raise ValueError("sample_bpr_batch: every sampled user needs at least one interaction")
    pos = indices[indptr[users] + (rng.random(batch_size) * deg).astype(np.int64)]
    neg = rng.integers(0, n_items, batch_size)
    for _ in range(64):  # rejection: resample the (few) negatives that hit an interacted item
        bad = np.array([n in indices[indptr[u]:indptr[u + 1]] for u, n in zip(users, neg)]) if batch_size <= 4096 else \
            np.zeros(batch_size, bool)
        if not bad.any():
            break
        neg[bad] = rng.integers(0, n_items, int(bad.sum()))
    return users.astype(np.int64), pos.astype(np.int64), neg.astype(np.int64)

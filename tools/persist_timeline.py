#!/usr/bin/env python3
"""Reads the records VISMA_ICP_PERSIST_TIMELINE=<file> makes a library write (one per persistent launch: for every pass
and workgroup the 100 MHz clock when the pass began -- its transform accepted -- and when its body, fold ticket
included, was done) and prints, per launch, the medians over the passes after the first of: how long the workgroups
take (median / slowest), when the last one is done after the first one began, and the gap from there to the first
workgroup's next begin (fold tail + statistics to the host + solve + command back + relay).
    VISMA_ICP_PERSIST_TIMELINE=/tmp/tl.bin python tools/persist_probe.py ... ; python tools/persist_timeline.py /tmp/tl.bin"""
import sys

import numpy as np


def main():
    raw = np.fromfile(sys.argv[1], dtype=np.uint64)
    pos, k = 0, 0
    while pos + 3 <= len(raw):
        passes, blocks, ran = (int(x) for x in raw[pos:pos + 3])
        n = 2 * passes * blocks
        rec64 = raw[pos + 3:pos + 3 + n].reshape(passes, blocks, 2)
        work = (rec64[:, :, 1] >> np.uint64(44)).astype(np.int64)          # queued queries | chunks << 12
        rec = rec64.copy()
        rec[:, :, 0] &= np.uint64(0xFFFFFFFFFFF)
        rec[:, :, 1] &= np.uint64(0xFFFFFFFFFFF)
        rec = rec.astype(np.int64)
        pos += 3 + n
        k += 1
        m = min(passes, ran)
        if m < 3:
            continue
        rec = rec[:m]
        t0 = rec[:, :, 0].min(axis=1)                     # first workgroup to begin the pass
        begin_spread = rec[:, :, 0].max(axis=1) - t0      # ... and the last
        body = rec[:, :, 1] - rec[:, :, 0]
        done = rec[:, :, 1].max(axis=1) - t0              # the slowest workgroup's end after the first begin
        gap = t0[1:] - rec[:-1, :, 1].max(axis=1)         # slowest end -> the next pass's first begin
        step = t0[1:] - t0[:-1]
        us = 0.01
        sel = slice(1, None)
        print("launch %d: %d passes x %d workgroups | step %.2f us | begin spread %.2f | body median %.2f / slowest %.2f | "
              "last done after first begin %.2f | slowest done -> next begin %.2f"
              % (k, m, blocks, np.median(step[sel]) * us if len(step) > 1 else float("nan"), np.median(begin_spread[sel]) * us,
                 np.median(np.median(body[sel], axis=1)) * us, np.median(body[sel].max(axis=1)) * us, np.median(done[sel]) * us,
                 np.median(gap[sel if len(gap) > 1 else slice(None)]) * us))
        # the distribution of the workgroups' bodies (the slowest is the one that finished the fold: it also summed and published)
        b = np.sort(body[sel], axis=1)
        q = lambda f: np.median(b[:, int(f * (blocks - 1))]) * us
        print("          body percentiles: 10%% %.2f  50%% %.2f  90%% %.2f  99%% %.2f  second slowest %.2f  slowest %.2f"
              % (q(0.1), q(0.5), q(0.9), q(0.99), np.median(b[:, -2]) * us if blocks > 1 else float("nan"), np.median(b[:, -1]) * us))
        # does a workgroup's time follow its work?  (last recorded pass)
        wq, wc, bt = work[m - 1] & 0xFFF, work[m - 1] >> 12, body[m - 1] * us
        order = np.argsort(bt)
        slow, rest = order[-max(blocks // 50, 1):], order[: -max(blocks // 50, 1)]
        print("          last pass: queued queries per workgroup mean %.1f max %d, chunks mean %.1f max %d | slowest 2%%: queries %.1f chunks %.1f "
              "body %.2f | the rest: queries %.1f chunks %.1f body %.2f | corr(body, chunks) %.2f"
              % (wq.mean(), wq.max(), wc.mean(), wc.max(), wq[slow].mean(), wc[slow].mean(), bt[slow].mean(), wq[rest].mean(),
                 wc[rest].mean(), bt[rest].mean(), np.corrcoef(bt, wc)[0, 1] if blocks > 2 and wc.std() > 0 else float("nan")))


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""LDS bank-conflict model of aa_conv_dw_frame_x6_kernel (csrc/conv_dw_frame_x6.hip), round 5.

Counts LDS cycles per workgroup for every LDS access of the kernel under these rules (they
reproduce the PMC counters of rounds 2-5 within ~10 %: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE):
  64 banks of 4 bytes; distinct dwords in one bank serialise;
  ds_read_b64_tr_b16  two passes of 32 lanes;   ds_write_b128 / ds_read_b128  four passes of 16 lanes;
  ds_write_b32        one pass of 64 lanes.
Prints the layout of rounds 2-4 and the round-5 one for the two DQN shapes:

  python tools/dw6_lds_model.py
"""
SEQ = [list(range(16 * i, 16 * i + 16)) for i in range(4)]


def cyc(addrs, width):
    banks = {}
    for a in addrs:
        if a is None:
            continue
        for d in range(width // 4):
            dw = a // 4 + d
            banks.setdefault(dw & 63, set()).add(dw)
    return max((len(v) for v in banks.values()), default=0)


def model(name, H, W, Cin, KH, KW, s, G, xpitch, xrow, zaddr, tile_pitch):
    OH, OW = (H - KH) // s + 1, (W - KW) // s + 1
    OHW = OH * OW
    KS, RTW = (OHW + 31) // 32, KW * Cin // 16 // 4
    xsh = (Cin // 8).bit_length() - 1
    tot, ideal = {}, {}

    def add(k, c, i):
        tot[k] = tot.get(k, 0) + c
        ideal[k] = ideal.get(k, 0) + i

    for wave in range(8):
        cp, rg = wave & 1, wave >> 1
        for ks in range(KS):
            for h in range(2):
                for half in range(2):
                    lanes = range(32 * half, 32 * half + 32)
                    pix = lambda l: 32 * ks + 16 * h + 4 * (l >> 4) + ((l & 15) >> 2)
                    for c in range(2):
                        ad = [zaddr(pix(l)) + (l & 3) * 8 + (cp * 2 + c) * 32 for l in lanes]
                        add("read dZ", 3 * cyc(ad, 8), 3)
                    for rt in range(RTW):
                        r0 = (rg * RTW + rt) * 16
                        kx, ci0 = divmod(r0, Cin)
                        ad = []
                        for l in lanes:
                            oy, ox = divmod(min(pix(l), OHW - 1), OW)
                            ad.append(oy * xrow + (ox * s + kx) * xpitch + ci0 * 2 + (l & 3) * 8)
                        add("read x", 3 * cyc(ad, 8), 3)
    nx, nz = OH * W * (Cin // 8), OHW * 8
    for u in range(2):
        for wave in range(8):
            for grp in SEQ:
                ad = []
                for l in grp:
                    it = wave * 64 + l + u * 512
                    if it >= nx:
                        ad.append(None)
                        continue
                    q, o = it >> xsh, it & ((1 << xsh) - 1)
                    row, xw = divmod(q, W)
                    ad.append(row * xrow + xw * xpitch + o * 16)
                if any(a is not None for a in ad):
                    add("stage x", 3 * cyc(ad, 16), 3)
                ad = []
                for l in grp:
                    it = wave * 64 + l + u * 512
                    ad.append(None if it >= nz else zaddr(it >> 3) + (it & 7) * 16)
                if any(a is not None for a in ad):
                    add("stage dZ", 3 * cyc(ad, 16), 3)
    tot = {k: v * G for k, v in tot.items()}
    ideal = {k: v * G for k, v in ideal.items()}
    for wave in range(8):
        cp, rg = wave & 1, wave >> 1
        for rt in range(RTW):
            for c in range(2):
                for e in range(4):
                    ad = [(((rg * RTW + rt) * 16 + 4 * (l >> 4) + e) * tile_pitch +
                           (cp * 2 + c) * 16 + (l & 15)) * 4 for l in range(64)]
                    add("slab tile", cyc(ad, 4), 1)
    T, I = sum(tot.values()), sum(ideal.values())
    print(f"{name}: " + ", ".join(f"{k} {tot[k]} ({tot[k] - ideal[k]} conflict)" for k in tot) +
          f"; conflict share {(T - I) / T:.3f}")


if __name__ == "__main__":
    for nm, (H, W, Cin, K, s, xp) in (("conv2.dW", (20, 20, 32, 4, 2, 80)),
                                      ("conv3.dW", (9, 9, 64, 3, 1, 160))):
        OW = (W - K) // s + 1
        model(nm + " rounds 2-4", H, W, Cin, K, K, s, 4, xp, W * xp, lambda p: p * 160, 80)
        xrow = W * xp
        while (xrow - OW * s * xp) % 256:
            xrow += 16
        model(nm + " round 5   ", H, W, Cin, K, K, s, 4, xp, xrow,
              lambda p: (p >> 1) * 288 + (p & 1) * 128, 68)

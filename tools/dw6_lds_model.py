import itertools, sys
GRP = [[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27],
       [4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31],
       [32,33,34,35,44,45,46,47,52,53,54,55,56,57,58,59],
       [36,37,38,39,40,41,42,43,48,49,50,51,60,61,62,63]]
SEQ = [list(range(16*i,16*i+16)) for i in range(4)]
def cyc(addrs_bytes, width):
    """addrs: list of byte addresses (one per lane in the pass), width bytes each"""
    banks = {}
    for a in addrs_bytes:
        if a is None: continue
        for d in range(width//4):
            dw = a//4 + d
            banks.setdefault(dw & 63, set()).add(dw)
    return max((len(v) for v in banks.values()), default=0)

def model(name, H, W, Cin, KH, KW, s, Cout, G, xpitch, zpitch, xrow=None, tile_pitch=80, b128grp=GRP, verbose=True):
    OH = (H-KH)//s+1; OW = (W-KW)//s+1; OHW = OH*OW
    KS = (OHW+31)//32; RTW = KW*Cin//16//4
    if xrow is None: xrow = W*xpitch
    xsh = (Cin//8).bit_length()-1; zsh = 3
    tot = {}; ideal = {}
    def add(k, c, i):
        tot[k] = tot.get(k,0)+c; ideal[k] = ideal.get(k,0)+i
    # tr reads (per frame), per wave: cp, rg
    for wave in range(8):
        cp, rg = wave & 1, wave >> 1
        for ks in range(KS):
            for h in range(2):
                for half in range(2):
                    lanes = range(32*half, 32*half+32)
                    # dz
                    for c in range(2):
                        ad = []
                        for l in lanes:
                            g, j = l>>4, l&15
                            p = 32*ks+16*h+4*g+(j>>2)
                            ad.append(p*zpitch + (j&3)*8 + (cp*2+c)*32)
                        add('tr_dz', 3*cyc(ad,8), 3)
                    for rt in range(RTW):
                        r0 = (rg*RTW+rt)*16; kx = r0//Cin; ci0 = r0-kx*Cin
                        toff = kx*xpitch+ci0*2
                        ad = []
                        for l in lanes:
                            g, j = l>>4, l&15
                            p = 32*ks+16*h+4*g+(j>>2)
                            pc = min(p, OHW-1)
                            oy, ox = divmod(pc, OW)
                            ad.append(oy*xrow + ox*s*xpitch + (j&3)*8 + toff)
                        add('tr_x', 3*cyc(ad,8), 3)
    # staging stores per frame (3 planes identical pattern modulo plane offset; plane offsets assumed multiple of 256? no: include)
    nx = OH*W*(Cin//8); nz = OHW*8
    for u in range(2):
        for wave in range(8):
            for grp in b128grp:
                ad = []
                for l in grp:
                    it = wave*64 + l + u*512
                    if it >= nx: ad.append(None); continue
                    q, o = it >> xsh, it & ((1<<xsh)-1)
                    row, xw = divmod(q, W)
                    ad.append(row*xrow + xw*xpitch + o*16)
                if any(a is not None for a in ad):
                    add('st_x', 3*cyc(ad,16), 3)
            for grp in b128grp:
                ad = []
                for l in grp:
                    it = wave*64 + l + u*512
                    if it >= nz: ad.append(None); continue
                    q, o = it >> 3, it & 7
                    ad.append(q*zpitch + o*16)
                if any(a is not None for a in ad):
                    add('st_z', 3*cyc(ad,16), 3)
    per_frame = dict(tot); per_frame_i = dict(ideal)
    tot = {k: v*G for k,v in tot.items()}; ideal = {k: v*G for k,v in ideal.items()}
    # epilogue tile writes (b32, all 64 lanes one pass) and reads b128
    for wave in range(8):
        cp, rg = wave&1, wave>>1
        for rt in range(RTW):
            for c in range(2):
                for e in range(4):
                    ad = []
                    for l in range(64):
                        g, j = l>>4, l&15
                        row = (rg*RTW+rt)*16+4*g+e
                        ad.append((row*tile_pitch + (cp*2+c)*16 + j)*4)
                    add('ep_w', cyc(ad,4), 1)
    rows_blk = KW*Cin
    n4 = rows_blk*16
    for q0 in range(0, n4, 64):
        for grp in b128grp:
            ad = []
            for l in grp:
                q = q0+l
                if q >= n4: ad.append(None); continue
                row, c4 = q>>4, q&15
                ad.append((row*tile_pitch + c4*4)*4)
            add('ep_r', cyc(ad,16), 1)
    T = sum(tot.values()); I = sum(ideal.values())
    if verbose:
        print(name, 'xpitch', xpitch, 'xrow', xrow, 'zpitch', zpitch, 'tile', tile_pitch)
        for k in tot: print('   %-6s cycles %6d ideal %6d conflict %6d' % (k, tot[k], ideal[k], tot[k]-ideal[k]))
        print('   total %d ideal %d conflict share %.3f' % (T, I, (T-I)/T))
    return T, I
if __name__ == '__main__':
    for grpname, g in (('perm', GRP), ('seq', SEQ)):
        print('== b128 groups:', grpname)
        model('conv2', 20,20,32,4,4,2,64,4, 80,160, b128grp=g)
        model('conv3', 9,9,64,3,3,1,64,4, 160,160, b128grp=g)

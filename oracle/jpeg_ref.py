"""Integer restatement of libjpeg(-turbo)'s baseline 4:2:0 JPEG round trip (TEST INFRASTRUCTURE, see oracle/__init__.py).

The reference's JPEG augmentation is a Pillow round trip (utils/image.py:13-34, valuemetric.py:21-50): Pillow is installed,
so ``pil_roundtrip`` IS the oracle.  ``jpeg_roundtrip`` restates what libjpeg does between the RGB bytes going in and the
RGB bytes coming out (jccolor.c fixed-point RGB->YCbCr, jcsample.c h2v2 box filter with alternating bias and its edge
padding rules, jfdctint.c jpeg_fdct_islow, jcdctmgr.c quantisation with the IJG quality tables, jidctint.c
jpeg_idct_islow, jdsample.c h2v2 fancy up-sampling, jdcolor.c YCbCr->RGB); entropy coding is lossless and omitted.
libjpeg sources are not available offline: the restatement is pinned by tests/test_oracle_aug.py, which checks it
BIT-EXACT against Pillow for many sizes / qualities.  The HIP kernel (csrc/aug.hip) is the same arithmetic.
"""
import io

import numpy as np
from PIL import Image


C = dict(F_0_298631336=2446, F_0_390180644=3196, F_0_541196100=4433, F_0_765366865=6270, F_0_899976223=7373, F_1_175875602=9633,
         F_1_501321110=12299, F_1_847759065=15137, F_1_961570560=16069, F_2_053119869=16819, F_2_562915447=20995, F_3_072711026=25172)
CONST_BITS, PASS1_BITS = 13, 2
def DESCALE(x, n): return (x + (1 << (n - 1))) >> n

BASE_L = np.array([16,11,10,16,24,40,51,61,12,12,14,19,26,58,60,55,14,13,16,24,40,57,69,56,14,17,22,29,51,87,80,62,18,22,37,56,68,109,103,77,24,35,55,64,81,104,113,92,49,64,78,87,103,121,120,101,72,92,95,98,112,100,103,99]).reshape(8,8)
BASE_C = np.array([17,18,24,47,99,99,99,99,18,21,26,66,99,99,99,99,24,26,56,99,99,99,99,99,47,66,99,99,99,99,99,99]+[99]*32).reshape(8,8)
def qtable(base, q):
    scale = 5000 // q if q < 50 else 200 - 2 * q
    t = (base * scale + 50) // 100
    return np.clip(t, 1, 255)

def fdct_1d(d, first):
    d = d.astype(np.int64)
    t0, t7 = d[...,0]+d[...,7], d[...,0]-d[...,7]
    t1, t6 = d[...,1]+d[...,6], d[...,1]-d[...,6]
    t2, t5 = d[...,2]+d[...,5], d[...,2]-d[...,5]
    t3, t4 = d[...,3]+d[...,4], d[...,3]-d[...,4]
    t10, t13, t11, t12 = t0+t3, t0-t3, t1+t2, t1-t2
    out = np.zeros_like(d)
    if first:
        out[...,0] = (t10+t11) << PASS1_BITS; out[...,4] = (t10-t11) << PASS1_BITS
        n = CONST_BITS-PASS1_BITS
    else:
        out[...,0] = DESCALE(t10+t11, PASS1_BITS); out[...,4] = DESCALE(t10-t11, PASS1_BITS)
        n = CONST_BITS+PASS1_BITS
    z1 = (t12+t13)*C['F_0_541196100']
    out[...,2] = DESCALE(z1 + t13*C['F_0_765366865'], n)
    out[...,6] = DESCALE(z1 - t12*C['F_1_847759065'], n)
    z1, z2, z3, z4 = t4+t7, t5+t6, t4+t6, t5+t7
    z5 = (z3+z4)*C['F_1_175875602']
    t4 = t4*C['F_0_298631336']; t5 = t5*C['F_2_053119869']; t6 = t6*C['F_3_072711026']; t7 = t7*C['F_1_501321110']
    z1 = -z1*C['F_0_899976223']; z2 = -z2*C['F_2_562915447']; z3 = -z3*C['F_1_961570560']; z4 = -z4*C['F_0_390180644']
    z3 = z3+z5; z4 = z4+z5
    out[...,7] = DESCALE(t4+z1+z3, n); out[...,5] = DESCALE(t5+z2+z4, n); out[...,3] = DESCALE(t6+z2+z3, n); out[...,1] = DESCALE(t7+z1+z4, n)
    return out

def idct_1d(c, first):
    c = c.astype(np.int64)
    z2, z3 = c[...,2], c[...,6]
    z1 = (z2+z3)*C['F_0_541196100']
    t2 = z1 - z3*C['F_1_847759065']; t3 = z1 + z2*C['F_0_765366865']
    z2, z3 = c[...,0], c[...,4]
    t0 = (z2+z3) << CONST_BITS; t1 = (z2-z3) << CONST_BITS
    t10, t13, t11, t12 = t0+t3, t0-t3, t1+t2, t1-t2
    t0, t1, t2, t3 = c[...,7], c[...,5], c[...,3], c[...,1]
    z1, z2, z3, z4 = t0+t3, t1+t2, t0+t2, t1+t3
    z5 = (z3+z4)*C['F_1_175875602']
    t0 = t0*C['F_0_298631336']; t1 = t1*C['F_2_053119869']; t2 = t2*C['F_3_072711026']; t3 = t3*C['F_1_501321110']
    z1 = -z1*C['F_0_899976223']; z2 = -z2*C['F_2_562915447']; z3 = -z3*C['F_1_961570560']; z4 = -z4*C['F_0_390180644']
    z3 = z3+z5; z4 = z4+z5
    t0 = t0+z1+z3; t1 = t1+z2+z4; t2 = t2+z2+z3; t3 = t3+z1+z4
    n = CONST_BITS-PASS1_BITS if first else CONST_BITS+PASS1_BITS+3
    out = np.zeros_like(c)
    out[...,0] = DESCALE(t10+t3, n); out[...,7] = DESCALE(t10-t3, n)
    out[...,1] = DESCALE(t11+t2, n); out[...,6] = DESCALE(t11-t2, n)
    out[...,2] = DESCALE(t12+t1, n); out[...,5] = DESCALE(t12-t1, n)
    out[...,3] = DESCALE(t13+t0, n); out[...,4] = DESCALE(t13-t0, n)
    return out

def codec_plane(p, qt):
    """p: uint8 [Hp,Wp] (multiples of 8) -> decoded uint8"""
    Hp, Wp = p.shape
    b = p.reshape(Hp//8, 8, Wp//8, 8).transpose(0,2,1,3).astype(np.int64) - 128   # [by,bx,row,col]
    d = fdct_1d(b, True)                       # rows (along col index)
    d = fdct_1d(d.transpose(0,1,3,2), False).transpose(0,1,3,2)   # columns
    q = qt.astype(np.int64) << 3
    a = np.abs(d)
    coef = np.sign(d) * ((a + (q >> 1)) // q)
    deq = coef * qt
    w = idct_1d(deq.transpose(0,1,3,2), True).transpose(0,1,3,2)   # columns first
    o = idct_1d(w, False)                     # rows
    o = np.clip(o + 128, 0, 255)
    return o.transpose(0,2,1,3).reshape(Hp, Wp).astype(np.uint8)

def jpeg_roundtrip(rgb, quality):
    """rgb uint8 [H,W,3] -> uint8 [H,W,3] emulating libjpeg baseline 4:2:0 encode + decode"""
    H, W, _ = rgb.shape
    Hp, Wp = (H + 15)//16*16, (W + 15)//16*16
    x = np.pad(rgb, ((0,Hp-H),(0,Wp-W),(0,0)), mode='edge').astype(np.int64)
    R, G, B = x[...,0], x[...,1], x[...,2]
    Y = (19595*R + 38470*G + 7471*B + 32768) >> 16
    Cb = (-11059*R - 21709*G + 32768*B + (128<<16) + 32767) >> 16
    Cr = (32768*R - 27439*G - 5329*B + (128<<16) + 32767) >> 16
    def down(p):
        s = p[0::2,0::2] + p[0::2,1::2] + p[1::2,0::2] + p[1::2,1::2]
        bias = np.tile(np.array([1,2]), s.shape[1]//2 + 1)[:s.shape[1]][None,:]
        return (s + bias) >> 2
    He = (H + 1) // 2                      # chroma rows that come from real (or 1-row replicated) pixels
    def padrows(p):                        # libjpeg pads the DOWNSAMPLED component to the iMCU height by replicating its last row
        p = p[:He]
        return np.vstack([p] + [p[-1:]] * (Hp // 2 - He))
    Cbd, Crd = padrows(down(Cb)), padrows(down(Cr))
    ql, qc = qtable(BASE_L, quality), qtable(BASE_C, quality)
    Yd = codec_plane(Y.astype(np.uint8), ql).astype(np.int64)
    Cbq = codec_plane(Cbd.astype(np.uint8), qc).astype(np.int64)
    Crq = codec_plane(Crd.astype(np.uint8), qc).astype(np.int64)
    ch, cw = (H+1)//2, (W+1)//2
    def up(p):
        p = p[:ch,:cw]
        pu = np.vstack([p[:1], p, p[-1:]])     # replicate top/bottom
        out = np.zeros((2*ch, 2*cw), dtype=np.int64)
        for v in range(2):
            near = pu[1:-1]
            far = pu[:-2] if v == 0 else pu[2:]
            cs = 3*near + far                    # column sums
            last = np.hstack([cs[:, :1], cs[:, :-1]])
            nxt = np.hstack([cs[:, 1:], cs[:, -1:]])
            e = (3*cs + last + 8) >> 4
            o = (3*cs + nxt + 7) >> 4
            e[:,0] = (cs[:,0]*4 + 8) >> 4
            o[:,-1] = (cs[:,-1]*4 + 7) >> 4
            out[v::2, 0::2] = e
            out[v::2, 1::2] = o
        return out[:H,:W]
    Cbu, Cru = up(Cbq), up(Crq)
    Yv = Yd[:H,:W]
    if cw <= 2:   # libjpeg uses the fancy up-sampler only when downsampled_width > 2
        Cbu = np.repeat(np.repeat(Cbq[:ch, :cw], 2, 0), 2, 1)[:H, :W]
        Cru = np.repeat(np.repeat(Crq[:ch, :cw], 2, 0), 2, 1)[:H, :W]
    cb, cr = Cbu - 128, Cru - 128
    Rr = Yv + ((91881*cr + 32768) >> 16)
    Bb = Yv + ((116130*cb + 32768) >> 16)
    Gg = Yv + ((-22554*cb + 32768 - 46802*cr) >> 16)
    return np.clip(np.stack([Rr,Gg,Bb],-1), 0, 255).astype(np.uint8)

def pil_roundtrip(rgb, q):
    buf = io.BytesIO(); Image.fromarray(rgb).save(buf, format='JPEG', quality=q); buf.seek(0)
    return np.asarray(Image.open(buf).convert('RGB'))


def pil_roundtrip(rgb: np.ndarray, quality: int) -> np.ndarray:
    """uint8 [H,W,3] -> Pillow JPEG encode/decode (the reference's own codec path)."""
    buf = io.BytesIO()
    Image.fromarray(rgb).save(buf, format="JPEG", quality=quality)
    buf.seek(0)
    return np.asarray(Image.open(buf).convert("RGB"))

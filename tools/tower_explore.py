"""exploration: tower basis of GF(2^128) over the LCH14 subfield, change-of-basis matrices, XOR counts"""
import sys, os, ctypes as C
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import oracle_lib as ol
from oracle_lib import elt, arr
o = ol.oracle()

def mul(a, b):
    r = o.lfo_gf_mul(elt(a), elt(b)); return (r.l[0], r.l[1])
def toint(a): return a[0] | (a[1] << 64)
def fromint(x): return (x & (2**64-1), x >> 64)

def minpoly(gi, m):
    """min poly of element (int) of degree m: find dependency among 1, g, ..., g^m"""
    pw = [1]
    for _ in range(m):
        pw.append(toint(mul(fromint(pw[-1]), fromint(gi))))
    # gaussian elimination on vectors pw[0..m] with tracking
    basis = {}  # pivot -> (vec, combo)
    for idx, v in enumerate(pw):
        combo = 1 << idx
        while v:
            p = v.bit_length() - 1
            if p in basis:
                bv, bc = basis[p]; v ^= bv; combo ^= bc
            else:
                basis[p] = (v, combo); break
        if v == 0:
            return combo  # bit j set => coefficient of x^j
    return None

def polyeval_in_field(mu, hi):
    """evaluate GF(2) polynomial mu at field element hi (int)"""
    acc = 0
    for j in range(mu.bit_length() - 1, -1, -1):
        acc = toint(mul(fromint(acc), fromint(hi)))
        if (mu >> j) & 1: acc ^= 1
    return acc


import time
def field_pow_table(gi, count):
    out=[1]
    for _ in range(count-1): out.append(toint(mul(fromint(out[-1]), fromint(gi))))
    return out

def rank_and_matrix(cols):
    """cols: list of 128 ints (basis vectors as 128-bit ints). returns rank"""
    basis={}
    r=0
    for v in cols:
        while v:
            p=v.bit_length()-1
            if p in basis: v^=basis[p]
            else: basis[p]=v; r+=1; break
    return r

def invert_bitmatrix(cols, nbits=128):
    """cols[j] = image of unit vector j (int).  Return list rows of inverse as ints: inv_cols[j] s.t. M*inv = I"""
    n=nbits
    # Solve by Gaussian elimination on augmented columns: treat matrix M with columns cols; want M^-1.
    # Work with rows: row i of M has bit j = (cols[j]>>i)&1
    rows=[0]*n
    for j,cv in enumerate(cols):
        for i in range(n):
            if (cv>>i)&1: rows[i]|=1<<j
    aug=[(rows[i], 1<<i) for i in range(n)]
    for col in range(n):
        piv=None
        for r in range(col,n):
            if (aug[r][0]>>col)&1: piv=r;break
        assert piv is not None
        aug[col],aug[piv]=aug[piv],aug[col]
        for r in range(n):
            if r!=col and (aug[r][0]>>col)&1:
                aug[r]=(aug[r][0]^aug[col][0], aug[r][1]^aug[col][1])
    # now aug[r][1] is row r of M^-1 (as bitmask over input bits)
    return [a[1] for a in aug]

def paar_xor_count(rows, nin):
    """greedy Paar CSE: rows = list of int masks over nin inputs. returns number of XOR gates"""
    rows=list(rows); nvars=nin; gates=0
    import itertools
    while True:
        # count pair frequencies
        best=0; bp=None
        # build column sets
        cnt={}
        for r in rows:
            bits=[i for i in range(nvars) if (r>>i)&1]
            if len(bits)<2: continue
            for a_i in range(len(bits)):
                for b_i in range(a_i+1,len(bits)):
                    key=(bits[a_i],bits[b_i]); cnt[key]=cnt.get(key,0)+1
        if not cnt: break
        bp=max(cnt,key=cnt.get); best=cnt[bp]
        if best<2: break
        a,b=bp; new=nvars; nvars+=1; gates+=1
        m_=(1<<a)|(1<<b)
        rows=[(r^m_)|(1<<new) if (r&m_)==m_ else r for r in rows]
    gates+=sum(max(0,bin(r).count("1")-1) for r in rows)
    return gates

for k,mu in ((4,0x1002B),(5,None)):
    c = ol.gf_ctx(k); m = 1<<k; d=128//m
    g = toint(tuple(int(x) for x in arr(c.g)))
    nu = minpoly(g, m)
    print("k=%d minpoly(g)=%#x weight %d" % (k, nu, bin(nu).count("1")))
    h=g; muse=nu
    if mu is not None:
        t0=time.time(); x=1
        for e in range(1,(1<<m)-1):
            x=toint(mul(fromint(x),fromint(g)))
            if e%2==1 or True:
                if polyeval_in_field(mu,x)==0:
                    h=x; muse=mu; print("found root of %#x: g^%d (%.1fs)"%(mu,e,time.time()-t0)); break
    # basis vectors h^j * X^i
    hp=field_pow_table(h,m)
    cols=[]
    for i in range(d):
        Xi=1<<i
        for j in range(m):
            cols.append(toint(mul(fromint(hp[j]),fromint(Xi))))
    print(" rank", rank_and_matrix(cols))
    inv=invert_bitmatrix(cols)   # rows of poly->tower map: tower bit p = parity(inv[p] & x)
    fwd_rows=[0]*128             # tower->poly: poly bit i = parity over tower bits p with (cols[p]>>i)&1
    for p_,cv in enumerate(cols):
        for i in range(128):
            if (cv>>i)&1: fwd_rows[i]|=1<<p_
    w_in=sum(bin(r).count("1") for r in inv); w_out=sum(bin(r).count("1") for r in fwd_rows)
    print(" poly->tower weight %d (naive XORs %d), tower->poly weight %d"%(w_in,w_in-128,w_out))
    t0=time.time(); print(" paar poly->tower:", paar_xor_count(inv,128), "tower->poly:", paar_xor_count(fwd_rows,128), "%.0fs"%(time.time()-t0))

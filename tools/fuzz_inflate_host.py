"""standalone fuzz driver for the host inflate stage (used by tools/asan_inflate_host.sh):
   python3 fuzz_inflate_host.py <lib.so> [seed] [cases]  -- the library must export zng_rocm_inflate_tokens_decode/_free"""
import ctypes as C, os, sys, zlib, random
import numpy as np
lib = C.CDLL(sys.argv[1])
class Tok(C.Structure):
    _fields_=[("tokens",C.c_void_p),("ntokens",C.c_size_t),("literals",C.c_void_p),("nliterals",C.c_size_t),
              ("segs",C.c_void_p),("nsegs",C.c_size_t),("out_len",C.c_uint64),("in_used",C.c_size_t),("status",C.c_int),("msg",C.c_char_p)]
lib.zng_rocm_inflate_tokens_decode.argtypes=[C.c_char_p,C.c_size_t,C.POINTER(Tok)]
lib.zng_rocm_inflate_tokens_free.argtypes=[C.POINTER(Tok)]
def replay(t):
    toks=np.ctypeslib.as_array(C.cast(t.tokens,C.POINTER(C.c_uint32)),(t.ntokens,)) if t.ntokens else np.zeros(0,np.uint32)
    lits=bytes((C.c_ubyte*t.nliterals).from_address(t.literals)) if t.nliterals else b""
    out=bytearray(); lp=0
    for tk in toks.tolist():
        if tk>>31:
            ln=((tk>>16)&0xff)+3; d=(tk&0xffff)+1
            for _ in range(ln): out.append(out[-d])
        else:
            out+=lits[lp:lp+tk]; lp+=tk
    return bytes(out)
rnd=random.Random(int(sys.argv[2]) if len(sys.argv)>2 else 1)
n_ok=n_err=0
if len(sys.argv)>4 and sys.argv[4]=="threads":
    # larger streams through the multi-threaded decode of ONE stream (block search, partial decodes, chain, join):
    # must agree with the one-thread decoder on status, bytes produced, input consumed -- and with CPython on the bytes
    lib.zng_rocm_inflate_tokens_decode_threads.argtypes=[C.c_char_p,C.c_size_t,C.c_uint32,C.c_int,C.POINTER(Tok)]
    nrng=np.random.default_rng(rnd.randrange(1<<30))
    words=[bytes(nrng.integers(97,123,size=int(k),dtype=np.uint8))+b" " for k in nrng.integers(1,12,size=3000)]
    for it in range(int(sys.argv[3])):
        pieces=[]
        for _ in range(rnd.randrange(2,6)):
            kind=rnd.randrange(3)
            if kind==0: pieces.append(b"".join(words[i] for i in (nrng.zipf(1.3,size=200000)-1)%len(words)))
            elif kind==1: pieces.append(bytes(nrng.integers(0,256,size=rnd.randrange(200000,900000),dtype=np.uint8)))
            else: pieces.append(bytes(nrng.integers(0,4,size=rnd.randrange(300000,2000000),dtype=np.uint8)))
        data=b"".join(pieces)
        c=zlib.compressobj(rnd.choice([1,6]),zlib.DEFLATED,-15)
        parts=[]
        for lo in range(0,len(data),1<<18):
            parts.append(c.compress(data[lo:lo+(1<<18)]))
            if rnd.randrange(4)==0: parts.append(c.flush(rnd.choice([zlib.Z_SYNC_FLUSH,zlib.Z_FULL_FLUSH])))
        comp=bytearray(b"".join(parts)+c.flush())
        mode=rnd.randrange(3)
        if mode==1:
            for _ in range(rnd.randrange(1,4)): comp[rnd.randrange(len(comp))]^=1<<rnd.randrange(8)
        elif mode==2: comp=comp[:rnd.randrange(len(comp)//2,len(comp))]
        comp=bytes(comp)
        a=Tok(); sa=lib.zng_rocm_inflate_tokens_decode(comp,len(comp),C.byref(a))
        b=Tok(); sb=lib.zng_rocm_inflate_tokens_decode_threads(comp,len(comp),0,rnd.choice([2,3,4,7]),C.byref(b))
        assert (sa,a.out_len,a.in_used)==(sb,b.out_len,b.in_used),(it,mode,sa,sb)
        if sb==1:
            d=zlib.decompressobj(-15); ref=d.decompress(comp)
            toks=np.ctypeslib.as_array(C.cast(b.tokens,C.POINTER(C.c_uint32)),(b.ntokens,))
            # replay with numpy where it can: literals vectorised, matches in Python
            assert d.eof and b.out_len==len(ref)
            got=replay(b) if b.ntokens<400000 else None
            assert got is None or got==ref,(it,mode)
            n_ok+=1
        else: n_err+=1
        lib.zng_rocm_inflate_tokens_free(C.byref(a)); lib.zng_rocm_inflate_tokens_free(C.byref(b))
    print("threads: ok",n_ok,"rejected",n_err)
    sys.exit(0)
for it in range(int(sys.argv[3]) if len(sys.argv)>3 else 400):
    kind=rnd.randrange(4)
    if kind==0: data=bytes(rnd.randrange(256) for _ in range(rnd.randrange(0,3000)))
    elif kind==1: data=(b"abc"*rnd.randrange(1,50)+bytes([rnd.randrange(256)]))*rnd.randrange(1,60)
    elif kind==2: data=bytes(rnd.choice(b"ab \n") for _ in range(rnd.randrange(0,5000)))
    else: data=os.urandom(rnd.randrange(1,200))*rnd.randrange(1,300)
    c=zlib.compressobj(rnd.choice([0,1,6,9]),zlib.DEFLATED,-15,8,rnd.choice([zlib.Z_DEFAULT_STRATEGY,zlib.Z_FIXED,zlib.Z_HUFFMAN_ONLY,zlib.Z_RLE]))
    comp=bytearray(c.compress(data)+c.flush())
    mode=rnd.randrange(4)
    if mode==1 and comp:
        for _ in range(rnd.randrange(1,4)): comp[rnd.randrange(len(comp))]^=1<<rnd.randrange(8)
    elif mode==2 and comp: comp=comp[:rnd.randrange(len(comp))]
    elif mode==3: comp=bytearray(os.urandom(rnd.randrange(1,400)))
    comp=bytes(comp)
    t=Tok(); st=lib.zng_rocm_inflate_tokens_decode(comp,len(comp),C.byref(t))
    try:
        d=zlib.decompressobj(-15); ref=d.decompress(comp); ref_ok=d.eof
    except zlib.error: ref=None; ref_ok=False
    if st==1:
        got=replay(t)
        assert ref_ok and got==ref and t.in_used==len(comp)-len(d.unused_data), (it,mode)
        n_ok+=1
    else:
        assert not ref_ok, (it,mode,st,t.msg)
        n_err+=1
    lib.zng_rocm_inflate_tokens_free(C.byref(t))
print("ok",n_ok,"rejected",n_err)

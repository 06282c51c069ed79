import sys, time, numpy as np, torch
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
from librir_amd import device as D
from librir_amd.synthetic import s1_noisy_background, inject_bad_pixels
from oracle.pyoracle import Oracle
O = Oracle()
print("device", D.device_available(), torch.cuda.get_device_name(0))
def chk(name, ok): print(("PASS " if ok else "FAIL ")+name); return ok
# --- codec small
for (n,h,w,gop) in [(5,67,83,3),(7,48,64,50),(3,20,20,2),(130,16,32,128),(4,64,64,50)]:
    rng = np.random.default_rng(n)
    fr = rng.integers(0,65536,(n,h,w)).astype(np.uint16) if n!=4 else np.full((n,h,w),1234,np.uint16)
    if n==7: fr = s1_noisy_background(n,h,w)
    if n==130: fr = (np.cumsum(np.ones((n,h,w),np.uint32),axis=2)+np.arange(n)[:,None,None]).astype(np.uint16)
    ctx = D.CodecContext(w,h,n,gop)
    t = torch.from_numpy(fr).cuda()
    enc = ctx.encode(t); dec = ctx.decode(enc); torch.cuda.synchronize()
    ok = np.array_equal(dec.cpu().numpy(), fr)
    # bitstream vs oracle per chunk
    L = ctx.layout
    hdr = enc.hdr.cpu().numpy().view(np.uint64); toff = enc.tile_off.cpu().numpy().view(np.uint32); coff = enc.chunk_off.cpu().numpy(); st = enc.stream.cpu().numpy().view(np.uint64)
    same = True
    for c in range(L.nchunks):
        f0 = c*gop; nf = min(gop, n-f0)
        h_o, o_o, st_o = O.codec_encode_chunk(fr[f0:f0+nf])
        same &= np.array_equal(hdr[c][:, :nf], h_o) and np.array_equal(toff[c], o_o) and np.array_equal(st[coff[c]:coff[c+1]], st_o) and (hdr[c][:, nf:]==0).all()
    chk(f"codec {n}x{h}x{w} gop{gop} roundtrip", ok); chk(f"codec {n}x{h}x{w} gop{gop} bitstream==oracle", same)
# --- codec full size timing
n,h,w=1000,512,640
fr = s1_noisy_background(n,h,w)
t = torch.from_numpy(fr).cuda()
ctx = D.CodecContext(w,h,n,50)
out = torch.empty_like(t)
for _ in range(2):
    enc = ctx.encode(t); ctx.decode(enc, out=out, check=False)
torch.cuda.synchronize()
e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True); e2=torch.cuda.Event(enable_timing=True)
e0.record(); enc = ctx.encode(t); e1.record(); ctx.decode(enc, out=out, check=False); e2.record(); torch.cuda.synchronize()
te, td = e0.elapsed_time(e1), e1.elapsed_time(e2)
print(f"encode {te:.3f} ms  decode {td:.3f} ms  fps {n/((te+td)/1e3):.0f}  raw GB/s {4*h*w*n/((te+td)/1e3)/1e9:.1f}  ratio {fr.nbytes/enc.compressed_bytes():.3f}")
chk("codec 1000 frames roundtrip", torch.equal(out.view(torch.int16), t.view(torch.int16)))
# --- filters
rng = np.random.default_rng(0)
for (h,w) in [(48,64),(67,83),(512,640)]:
    img16 = s1_noisy_background(3,h,w); t16 = torch.from_numpy(img16).cuda()
    for strat in ["", "background","wrap","nearest"]:
        for (dx,dy) in [(0,0),(1.25,-2.5),(-0.75,0.1),(w+1,0),(0.5,0.25)]:
            g = D.translate(t16,(dx,dy),strat,background=7).cpu().numpy()
            r = np.stack([O.translate(img16[i],dx,dy,strat,background=7) for i in range(3)])
            if not np.array_equal(g,r): chk(f"translate u16 {h}x{w} {strat} {dx},{dy} ndiff={(g!=r).sum()}", False)
    f32 = (rng.random((2,h,w))*1000).astype(np.float32); tf = torch.from_numpy(f32).cuda()
    g = D.translate(tf,(1.25,-2.5),"nearest").cpu().numpy(); r = np.stack([O.translate(f32[i],1.25,-2.5,"nearest") for i in range(2)])
    chk(f"translate f32 {h}x{w}", np.array_equal(g,r))
    for s in [0.5,0.75,1.0,2.0]:
        g = D.gaussian_filter(tf,s).cpu().numpy(); r = np.stack([O.gaussian_filter(f32[i],s) for i in range(2)])
        chk(f"gaussian {h}x{w} s={s} bitexact={np.array_equal(g,r)} maxrel={np.abs(g-r).max()/np.abs(r).max():.2e}", np.allclose(g,r,rtol=1e-5,atol=0))
    bad = inject_bad_pixels(img16, max(2,(h*w)//1600))
    tb = torch.from_numpy(bad).cuda()
    bp = D.BadPixels(tb[0])
    xy_o = O.bad_pixels_detect(bad[0]); fd, fc = O.bad_pixels_stats(bad[0])
    chk(f"badpix detect {h}x{w} n={bp.count}", np.array_equal(bp.positions(), xy_o) and bp.floor_correct==fc and bp.floor_detect==fd)
    g = bp.correct(tb).cpu().numpy(); r = np.stack([O.bad_pixels_correct(bad[i], xy_o, fc) for i in range(3)])
    chk(f"badpix correct {h}x{w}", np.array_equal(g,r))
    bp2 = D.BadPixels(tb[0], rows=h-3)
    tb2 = tb.clone(); bp2.remove_inplace(tb2, h-3)
    xy2 = O.bad_pixels_detect(bad[0][:h-3])
    r = np.stack([O.remove_bad_pixels(bad[i], xy2, rows=h-3) for i in range(3)])
    chk(f"remove_bad_pixels {h}x{w}", np.array_equal(tb2.cpu().numpy(), r) and np.array_equal(bp2.positions(), xy2))
    sh = np.array([[1.25,-2.5],[0,0],[-3.5,4.75]],np.float32)
    g = D.remove_motion(t16, sh, rows=h-3).cpu().numpy(); r = np.stack([O.remove_motion(img16[i], sh[i,0], sh[i,1], rows=h-3) for i in range(3)])
    chk(f"remove_motion {h}x{w}", np.array_equal(g,r))
    for p in [0,0.2,0.5,0.99,1.0]:
        g = D.find_median_pixel(t16,p).cpu().numpy(); r = [O.find_median_pixel(img16[i],p) for i in range(3)]
        m = (rng.random((3,h,w))<0.3).astype(np.uint8)
        g2 = D.find_median_pixel(t16,p,torch.from_numpy(m).cuda()).cpu().numpy(); r2 = [O.find_median_pixel(img16[i],p,m[i]) for i in range(3)]
        if not (list(g)==r and list(g2)==r2): chk(f"median pixel {h}x{w} p={p} {g} {r} {g2} {r2}", False)
    chk(f"median filter {h}x{w}", np.array_equal(D.median_filter(t16).cpu().numpy(), np.stack([O.median_filter(img16[i]) for i in range(3)])))
print("done")

"""numpy model of conv_wgrad_b3.hip's index logic — staging slots, LDS entry layout, per-column-shift
copies, tap bases, fragment reads (`base + lane`) and the wave -> output tile map — with values kept in
fp64 (no bf16 splitting), checked against a direct evaluation of the weight gradient. Written BEFORE the
kernel ran on the GPU (it was correct at first launch); re-run it after any change to the layout:
    python tools/exp/wgrad_b3_layout_model.py
(The tile-row choice mirrors pg_wgrad_b3_launch with the 3-slot caps of the first version.)"""
import numpy as np
def run(N,Cin,Cout,H,W,taps,seed=0):
    rng=np.random.default_rng(seed)
    x=rng.standard_normal((N,Cin,H,W)); dy=rng.standard_normal((N,Cout,H,W))
    T=len(taps); tap_dr=[t[0] for t in taps]; tap_dc=[t[1] for t in taps]
    min_dr=min(tap_dr); max_dr=max(tap_dr); hr=max_dr-min_dr
    dcs=[]; copy_of=[]
    for t in range(T):
        if tap_dc[t] not in dcs: dcs.append(tap_dc[t])
        copy_of.append(dcs.index(tap_dc[t]))
    ndc=len(dcs); PBR=W//8
    TR=0
    for tr in range(1,H+4):
        if (tr*PBR)%4: continue
        ds=64*tr*PBR; xs=32*(tr+hr)*PBR
        if ds>3*256 or xs>3*256 or (3*ds+3*ndc*xs)*16>76*1024: break
        TR=tr
        if tr>=H: break
    assert TR>0
    xh=TR+hr; tpi=(H+TR-1)//TR; total=N*tpi; dpb=TR*PBR; xpb=xh*PBR; ksteps=dpb//4
    dslots=64*dpb; xslots=32*xpb
    # single "piece": planes
    dplane=4*dpb*16; xplane=2*xpb*16
    x_off=dplane  # one piece only in the emulation
    tap_base=[copy_of[t]*xplane+(tap_dr[t]-min_dr)*PBR*16 for t in range(T)]
    dw=np.zeros((Cout,Cin,T))
    for co0 in range(0,Cout,64):
      for ci0 in range(0,Cin,32):
        for tile in range(total):
            n=tile//tpi; row0=(tile-n*tpi)*TR
            lds=np.full((x_off+ndc*xplane,8),np.nan)
            for e0 in range(dslots):
                e=e0; i=e&15; e>>=4; cb=e%PBR; e//=PBR; tr=e%TR; cot=e//TR
                ent=(cot*dpb+tr*PBR+cb)*16+i
                r=row0+tr
                lds[ent]=dy[n,co0+cot*16+i,r,8*cb:8*cb+8] if r<H else 0
            for e0 in range(xslots):
                e=e0; i=e&15; e>>=4; cb=e%PBR; e//=PBR; tr=e%xh; cit=e//xh
                ent=x_off+(cit*xpb+tr*PBR+cb)*16+i
                ir=row0+min_dr+tr
                ok=0<=ir<H
                ch=ci0+cit*16+i
                v8=x[n,ch,ir,8*cb:8*cb+8] if ok else np.zeros(8)
                m1=x[n,ch,ir,8*cb-1] if ok and cb>0 else 0.0
                p1=x[n,ch,ir,8*cb+8] if ok and cb<PBR-1 else 0.0
                for v in range(ndc):
                    dc=dcs[v]
                    if dc==0: val=v8
                    elif dc<0: val=np.concatenate(([m1],v8[:7]))
                    else: val=np.concatenate((v8[1:],[p1]))
                    lds[ent+v*xplane]=val
            assert not np.isnan(lds).any()
            for wave in range(4):
                wc=wave&1; wi=wave>>1
                for ks in range(ksteps):
                    for m in range(2):
                        A=np.zeros((16,32))
                        for lane in range(64):
                            A[lane&15,(lane>>4)*8:(lane>>4)*8+8]=lds[2*wc*dpb*16+m*dpb*16+ks*64+lane]
                        for t in range(T):
                            B=np.zeros((32,16))
                            for lane in range(64):
                                B[(lane>>4)*8:(lane>>4)*8+8,lane&15]=lds[x_off+tap_base[t]+wi*xpb*16+ks*64+lane]
                            D=A@B
                            cs=co0+(2*wc+m)*16; cis=ci0+wi*16
                            dw[cs:cs+16,cis:cis+16,t]+=D
    # reference
    ref=np.zeros((Cout,Cin,T))
    xp=np.pad(x,((0,0),(0,0),(2,2),(2,2)))
    for t,(dr,dc) in enumerate(taps):
        xs=xp[:,:,2+dr:2+dr+H,2+dc:2+dc+W]
        ref[:,:,t]=np.einsum('nohw,nihw->oi',dy,xs)
    err=np.abs(dw-ref).max()/np.abs(ref).max()
    print(N,Cin,Cout,H,W,taps,'TR',TR,'err',err)
    assert err<1e-12
run(2,32,64,8,8,[(-1,-1),(-1,0),(0,-1),(0,0)])
run(1,64,64,6,16,[(-1,-1),(-1,0),(0,-1),(0,0)])
run(1,32,128,5,32,[(0,-1),(0,0),(0,1)])
run(2,32,64,4,8,[(-1,0),(0,0)])
run(1,32,64,8,16,[(dr,dc) for dr in (-1,0,1) for dc in (-1,0,1)])
run(1,32,64,8,8,[(dr,dc) for dr in (-1,0) for dc in (-1,0,1)])

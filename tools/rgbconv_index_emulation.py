"""Index-level emulation of csrc/rgbconv.hip (fp64 arithmetic, the kernel's index formulas transcribed) against torch convs."""
import numpy as np, torch, torch.nn.functional as TF
import sys
torch.manual_seed(1)

def mma16(A, Bm, D):
    # per-lane fragments: A[lane][e] = A[i=l&15][k=4*(l>>4)+e]; B[lane][e] = B[k=4*(l>>4)+e][j=l&15]; D[lane][r] = D[i=4*(l>>4)+r][j=l&15]
    Am = np.zeros((16,16)); Bmm = np.zeros((16,16))
    for l in range(64):
        for e in range(4):
            Am[l&15, 4*(l>>4)+e] = A[l][e]
            Bmm[4*(l>>4)+e, l&15] = Bm[l][e]
    Dm = Am @ Bmm
    out = [list(D[l]) for l in range(64)]
    for l in range(64):
        for r in range(4):
            out[l][r] += Dm[4*(l>>4)+r, l&15]
    return out

def pack(w0, s0, wr, sr, br, C):
    wf = np.zeros(3*C*16); wd = np.zeros(9*16*C)
    for e in range(3*C*16):
        ky = e // (C*16); o = (e//16) % C; k = e % 16; kx = k>>2; j = k&3
        v = 0.0
        if kx < 3:
            for i in range(C):
                w = w0[((o*C+i)*3+ky)*3+kx]
                v += w * (wr[i*3+j] if j<3 else br[i])
            v *= s0*sr if j<3 else s0
        wf[e] = v
    for e in range(9*16*C):
        tap = e // (16*C); j = (e//C) % 16; o = e % C; ky = 2 - tap//3; kx = 2 - tap%3
        v = 0.0
        if j < 3:
            for i in range(C): v += w0[((o*C+i)*3+ky)*3+kx]*wr[i*3+j]
            v *= s0*sr
        wd[e] = v
    return wf, wd

def fwd(img, wf, b0, B, H, W, CB, EPI, ones):
    C=16*CB; R = 2 if EPI else 1
    TH = (16 if CB==1 else 8) if EPI else 16; TW=64
    IH=TH+2*R; IW=TW+2*R; ZH=TH+2*(R-1); ZW=TW+2*(R-1); NPX=ZH*ZW; NG=(NPX+15)//16; IPX=IH*IW+4
    tiles_x=W//TW; tiles_y=H//TH
    y = np.zeros((B,H,W,C)); bits = np.zeros((B,H,W,C//8),dtype=np.uint8)
    for blk in range(B*tiles_x*tiles_y):
        t=blk; tx=t%tiles_x; t//=tiles_x; ty=t%tiles_y; b=t//tiles_y
        ty0=ty*TH; tx0=tx*TW
        imgl = np.zeros((IPX,4))
        for idx in range(IPX):
            r=idx//IW; c=idx-r*IW; gy=ty0-R+r; gx=tx0-R+c
            ok = idx<IH*IW and 0<=gy<H and 0<=gx<W
            if ok: imgl[idx,:3]=img[b,gy,gx]; imgl[idx,3]=1.0 if ones else 0.0
        zl = np.zeros((NPX,C))
        for wave in range(4):
            wfr = [[[ [wf[(ky*C+cb*16+(l&15))*16+4*(l>>4)+e] for e in range(4)] for l in range(64)] for ky in range(3)] for cb in range(CB)]
            for g in range(wave, NG, 4):
                acc=[[ [0.0]*4 for l in range(64)] for cb in range(CB)]
                meta=[]
                for l in range(64):
                    px=g*16+(l&15); pxc=min(px,NPX-1); zr=pxc//ZW; zc=pxc-zr*ZW; meta.append((px,zr,zc))
                for ky in range(3):
                    bf=[list(imgl[(meta[l][1]+ky)*IW+meta[l][2]+(l>>4)]) for l in range(64)]
                    for cb in range(CB): acc[cb]=mma16(wfr[cb][ky], bf, acc[cb])
                for l in range(64):
                    px,zr,zc=meta[l]; l4=l>>4
                    gy=ty0-(R-1)+zr; gx=tx0-(R-1)+zc
                    inimg = px<NPX and 0<=gy<H and 0<=gx<W
                    for cb in range(CB):
                        if EPI==0:
                            if inimg: y[b,gy,gx,cb*16+4*l4:cb*16+4*l4+4]=acc[cb][l]
                        else:
                            bb=b0[cb*16+4*l4:cb*16+4*l4+4]
                            a=[ (lambda v: v if v>0 else 0.2*v)(acc[cb][l][r]+bb[r]) if inimg else 0.0 for r in range(4)]
                            if px<NPX: zl[px,cb*16+4*l4:cb*16+4*l4+4]=a
        if EPI==1:
            VPP=C//8; NSTRIP=TW*VPP; RSPLIT=256//NSTRIP; RPT=TH//RSPLIT
            for tid in range(256):
                s=tid%NSTRIP; half=tid//NSTRIP; c=s//VPP; v=s%VPP; r0=half*RPT
                h0=np.zeros(8); h1=np.zeros(8); cprev=np.zeros(8)
                for rr in range(RPT+2):
                    base=(r0+rr)*ZW+c
                    L=zl[base,v*8:v*8+8]; M=zl[base+1,v*8:v*8+8]; Rr=zl[base+2,v*8:v*8+8]
                    h=L+2*M+Rr
                    if rr>=2:
                        gy=ty0+r0+rr-2; gx=tx0+c
                        if gy<H and gx<W:
                            y[b,gy,gx,v*8:v*8+8]=(h0+2*h1+h)*0.0625
                            bb=0
                            for q in range(8):
                                if cprev[q]>0: bb|=1<<q
                            bits[b,gy,gx,v]=bb
                    h0=h1.copy(); h1=h.copy(); cprev=M.copy()
    return y,bits

def dgrad(gz, wd, B,H,W,CB):
    C=16*CB; TH=16 if CB==1 else 8; TW=64; GH=TH+2; GW=TW+2; VPP=C//8
    tiles_x=W//TW; tiles_y=H//TH
    gi=np.zeros((B,H,W,3))
    for blk in range(B*tiles_x*tiles_y):
        t=blk; tx=t%tiles_x; t//=tiles_x; ty=t%tiles_y; b=t//tiles_y; ty0=ty*TH; tx0=tx*TW
        gl=np.zeros((GH*GW,C))
        for idx in range(GH*GW*VPP):
            p=idx//VPP; v=idx-p*VPP; r=p//GW; c=p-r*GW; gy=ty0-1+r; gx=tx0-1+c
            if 0<=gy<H and 0<=gx<W: gl[p,v*8:v*8+8]=gz[b,gy,gx,v*8:v*8+8]
        for wave in range(4):
            wfr=[[ [[wd[(tap*16+(l&15))*C+cb*16+4*(l>>4)+e] for e in range(4)] for l in range(64)] for cb in range(CB)] for tap in range(9)]
            for g in range(wave, TH*4, 4):
                r=g>>2
                acc=[[0.0]*4 for l in range(64)]
                for ky in range(3):
                    for kx in range(3):
                        for cb in range(CB):
                            bf=[list(gl[(r+ky)*GW+((g&3)*16+(l&15))+kx, cb*16+4*(l>>4):cb*16+4*(l>>4)+4]) for l in range(64)]
                            acc=mma16(wfr[ky*3+kx][cb], bf, acc)
                for l in range(64):
                    c=(g&3)*16+(l&15); gy=ty0+r; gx=tx0+c
                    if (l>>4)==0 and gy<H and gx<W: gi[b,gy,gx]=acc[l][:3]
    return gi

def wgrad(img, gz, B,H,W,CB,ones,nblk):
    C=16*CB; TH=8; TW=64; TP=TH*TW; RH=TH+2; VPP=C//8
    tiles_x=W//TW; tiles_y=H//TH; ntiles=B*tiles_x*tiles_y
    NOUT=C*48
    part=np.zeros((nblk,NOUT))
    for blk in range(nblk):
        accs=[[[ [0.0]*4 for l in range(64)] for ky in range(3)] for cb in range(CB)]
        accs=[accs for wave in range(4)]
        import copy
        accs=[copy.deepcopy(accs[0]) for _ in range(4)]
        for t in range(blk, ntiles, nblk):
            tt=t; tx=tt%tiles_x; tt//=tiles_x; ty=tt%tiles_y; b=tt//tiles_y; ty0=ty*TH; tx0=tx*TW
            TPS=TP+8; IPS=RH*TW+8; gzT=np.zeros(C*TPS); imgT=np.zeros(12*IPS)
            for idx in range(TP*VPP):
                p=idx//VPP; v=idx-p*VPP; r=p//TW; c=p-r*TW; gy=ty0+r; gx=tx0+c
                val = gz[b,gy,gx,v*8:v*8+8] if (gy<H and gx<W) else np.zeros(8)
                for q in range(4):
                    gzT[(v*8+2*q)*TPS+p]=val[2*q]; gzT[(v*8+2*q+1)*TPS+p]=val[2*q+1]
            NIP=RH*(TW+2)
            for idx in range(NIP):
                rr=idx//(TW+2); cr=idx-rr*(TW+2); gy=ty0-1+rr; gx=tx0-1+cr
                ok=0<=gy<H and 0<=gx<W
                vj=[0,0,0,0]
                if ok: vj=list(img[b,gy,gx])+[1.0 if ones else 0.0]
                for kx in range(3):
                    c=cr-kx
                    if 0<=c<TW:
                        for j in range(4): imgT[(kx*4+j)*IPS+rr*TW+c]=vj[j]
            for wave in range(4):
                for g in range(wave, TH*4, 4):
                    r=g>>2
                    af=[[ [gzT[(cb*16+(l&15))*TPS+r*TW+(g&3)*16+4*(l>>4)+e] for e in range(4)] for l in range(64)] for cb in range(CB)]
                    for ky in range(3):
                        bf=[]
                        for l in range(64):
                            l15=l&15; nkx=(l15>>2) if (l15>>2)<3 else 0; nj=l15&3
                            c0=(g&3)*16+4*(l>>4)
                            bf.append([imgT[(nkx*4+nj)*IPS+(r+ky)*TW+c0+e] for e in range(4)])
                        for cb in range(CB): accs[wave][cb][ky]=mma16(af[cb], bf, accs[wave][cb][ky])
        for cb in range(CB):
            for ky in range(3):
                for rg in range(4):
                    for ln in range(64):
                        s=sum(accs[w][cb][ky][ln][rg] for w in range(4))
                        o=cb*16+4*(ln>>4)+rg; n=ln&15
                        part[blk,(o*3+ky)*16+n]=s
    return part.sum(0)

def chain(dwp,w0,s0,wr,sr,br,C):
    dw0=np.zeros(C*C*9); db0=np.zeros(C); dwr=np.zeros(C*3); dbr=np.zeros(C)
    for e in range(C*C*9):
        o=e//(C*9); i=(e//9)%C; tap=e%9; ky=tap//3; kx=tap%3
        d=dwp[(o*3+ky)*16+kx*4:(o*3+ky)*16+kx*4+4]
        v=s0*sr*(d[0]*wr[i*3]+d[1]*wr[i*3+1]+d[2]*wr[i*3+2]) + s0*d[3]*br[i]
        dw0[e]=v
    for o in range(C): db0[o]=dwp[(o*3+1)*16+4+3]
    for e in range(C*4):
        i=e>>2; j=e&3; v=0.0
        for o in range(C):
            for tap in range(9): v+=dwp[(o*3+tap//3)*16+(tap%3)*4+j]*w0[(o*C+i)*9+tap]
        v*= s0*sr if j<3 else s0
        if j<3: dwr[i*3+j]=v
        else: dbr[i]=v
    return dw0,db0,dwr,dbr

if __name__=="__main__":
    CB=int(sys.argv[1]) if len(sys.argv)>1 else 1
    C=16*CB; B=int(sys.argv[2]) if len(sys.argv)>2 else 1; H=int(sys.argv[3]) if len(sys.argv)>3 else 16; W=int(sys.argv[4]) if len(sys.argv)>4 else 64
    w0=torch.randn(C,C,3,3,dtype=torch.float64,requires_grad=True); b0=torch.randn(C,dtype=torch.float64,requires_grad=True)
    wr=torch.randn(C,3,1,1,dtype=torch.float64,requires_grad=True); br=torch.randn(C,dtype=torch.float64,requires_grad=True)
    s0,sr=0.3,0.7
    img=torch.randn(B,3,H,W,dtype=torch.float64)
    f=TF.conv2d(img, wr*sr, br); z=TF.conv2d(f, w0*s0, b0, padding=1)
    a=TF.leaky_relu(z,0.2); k=torch.tensor([1.,2.,1.],dtype=torch.float64); k=(k[:,None]*k[None,:]/16)[None,None].expand(C,1,3,3)
    xb=TF.conv2d(a,k,padding=1,groups=C)
    wf,wd=pack(w0.detach().numpy().ravel(),s0,wr.detach().numpy().ravel(),sr,br.detach().numpy(),C)
    imgn=img.permute(0,2,3,1).numpy()
    y,bits=fwd(imgn,wf,b0.detach().numpy(),B,H,W,CB,1,1)
    print("fwd epi1 err", np.abs(y-xb.detach().permute(0,2,3,1).numpy()).max())
    want_bits=(z.detach().permute(0,2,3,1).numpy()>0)
    got=np.unpackbits(bits[...,None],axis=-1,bitorder="little").reshape(B,H,W,C).astype(bool)
    print("bits mismatches", (got!=want_bits).sum())
    f2=TF.conv2d(img, wr*sr); z2=TF.conv2d(f2, w0*s0, padding=1)
    y0,_=fwd(imgn,wf,None,B,H,W,CB,0,0)
    print("fwd epi0 err", np.abs(y0-z2.detach().permute(0,2,3,1).numpy()).max())
    gz=torch.randn(B,C,H,W,dtype=torch.float64)
    imgr=img.clone().requires_grad_(True)
    z3=TF.conv2d(TF.conv2d(imgr, wr*sr), w0*s0, padding=1)
    (gi,)=torch.autograd.grad((z3*gz).sum(), imgr)
    gzn=gz.permute(0,2,3,1).numpy()
    gi2=dgrad(gzn,wd,B,H,W,CB)
    print("dgrad err", np.abs(gi2-gi.permute(0,2,3,1).numpy()).max())
    want=torch.autograd.grad((z*gz).sum(), [w0,b0,wr,br])
    dwp=wgrad(imgn,gzn,B,H,W,CB,1,3)
    dw0,db0,dwr,dbr=chain(dwp,w0.detach().numpy().ravel(),s0,wr.detach().numpy().ravel(),sr,br.detach().numpy(),C)
    for a_,b_,n in ((dw0,want[0],'w0'),(db0,want[1],'b0'),(dwr,want[2],'wr'),(dbr,want[3],'br')):
        print("wgrad", n, np.abs(a_-b_.numpy().ravel()).max()/np.abs(b_.numpy()).max())

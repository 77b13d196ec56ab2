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
    wd = np.zeros(3*16*C)
    for e in range(3*16*C):
        kyp = e // (16*C); i = (e//C) % 16; o = e % C; kxp = i>>2; j = i&3; ky = 2-kyp; kx = 2-kxp
        v = 0.0
        if j < 3 and kxp < 3:
            for ii in range(C): v += w0[((o*C+ii)*3+ky)*3+kx]*wr[ii*3+j]
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

RC_STRIP=14; RC_ROWS=32
def lrelu(v): return v if v>0 else 0.2*v
def fwdblur(img, wf, b0, B,H,W,CB,ones):
    C=16*CB; nstrips=(W+RC_STRIP-1)//RC_STRIP; nrb=(H+RC_ROWS-1)//RC_ROWS
    y=np.zeros((B,H,W,C)); bits=np.zeros((B,H,W,C//8),dtype=np.uint8)
    for item0 in range(B*nstrips*nrb):
        item=item0; sx=item%nstrips; item//=nstrips; rbk=item%nrb; b=item//nrb
        r_begin=rbk*RC_ROWS; r_end=min(r_begin+RC_ROWS,H)
        L=range(64)
        zc=[sx*RC_STRIP-1+(l&15) for l in L]; pc=[zc[l]-1+(l>>4) for l in L]
        pc_ok=[(l>>4)<3 and 0<=pc[l]<W for l in L]
        wfr=[[ [[wf[(ky*C+cb*16+(l&15))*16+4*(l>>4)+e] for e in range(4)] for l in L] for ky in range(3)] for cb in range(CB)]
        def load_row(gy): return [ (list(img[b,gy,pc[l]]) if (pc_ok[l] and 0<=gy<H) else [0.,0.,0.]) for l in L]
        def frag_of(v,gy): return [ v[l]+[1.0 if (pc_ok[l] and 0<=gy<H and ones) else 0.0] for l in L]
        zrow=r_begin-1
        f0=frag_of(load_row(zrow-1),zrow-1); f1=frag_of(load_row(zrow),zrow)
        n0=load_row(zrow+1); n1=load_row(zrow+2); n2=load_row(zrow+3)
        h1=[[[0.]*4 for l in L] for cb in range(CB)]; h2=[[[0.]*4 for l in L] for cb in range(CB)]; a1=[[[0.]*4 for l in L] for cb in range(CB)]
        while zrow<=r_end:
            f2=frag_of(n0,zrow+1); n0=n1; n1=n2; n2=load_row(zrow+4)
            for cb in range(CB):
                acc=[[0.]*4 for l in L]
                acc=mma16(wfr[cb][0],f0,acc); acc=mma16(wfr[cb][1],f1,acc); acc=mma16(wfr[cb][2],f2,acc)
                a=[[ (lrelu(acc[l][i]+b0[cb*16+4*(l>>4)+i]) if (0<=zc[l]<W and 0<=zrow<H) else 0.0) for i in range(4)] for l in L]
                def shr(l,i): return a[l-1][i] if (l&15)>0 else 0.0
                def shl(l,i): return a[l+1][i] if (l&15)<15 else 0.0
                h=[[shr(l,i)+2*a[l][i]+shl(l,i) for i in range(4)] for l in L]
                orow=zrow-1
                if orow>=r_begin:
                    for l in L:
                        l15=l&15; l4=l>>4
                        col_out = 1<=l15<=RC_STRIP and zc[l]<W
                        if col_out:
                            for i in range(4): y[b,orow,zc[l],cb*16+4*l4+i]=(h2[cb][l][i]+2*h1[cb][l][i]+h[l][i])*0.0625
                        nib=sum((1<<i) for i in range(4) if a1[cb][l][i]>0)
                        lo=l^16; other=sum((1<<i) for i in range(4) if a1[cb][lo][i]>0)
                        if col_out and not (l4&1): bits[b,orow,zc[l],cb*2+(l4>>1)]=nib|(other<<4)
                h2[cb]=h1[cb]; h1[cb]=h; a1[cb]=a
            f0=f1; f1=f2; zrow+=1
    return y,bits

def dgrad(gz, wd, B,H,W,CB):
    C=16*CB; nstrips=(W+RC_STRIP-1)//RC_STRIP; nrb=(H+RC_ROWS-1)//RC_ROWS
    gi=np.zeros((B,H,W,3))
    for item0 in range(B*nstrips*nrb):
        item=item0; sx=item%nstrips; item//=nstrips; rbk=item%nrb; b=item//nrb
        r_begin=rbk*RC_ROWS; r_end=min(r_begin+RC_ROWS,H)
        L=range(64)
        pc=[sx*RC_STRIP-1+(l&15) for l in L]; pc_ok=[0<=pc[l]<W for l in L]
        wfr=[[ [[wd[(ky*16+(l&15))*C+cb*16+4*(l>>4)+e] for e in range(4)] for l in L] for cb in range(CB)] for ky in range(3)]
        def load_row(gy): return [[ (list(gz[b,gy,pc[l],cb*16+4*(l>>4):cb*16+4*(l>>4)+4]) if (pc_ok[l] and 0<=gy<H) else [0.]*4) for l in L] for cb in range(CB)]
        r=r_begin
        g0=load_row(r-1); g1=load_row(r); n0=load_row(r+1); n1=load_row(r+2); n2=load_row(r+3)
        while r<r_end:
            g2=n0; n0=n1; n1=n2; n2=load_row(r+4)
            acc=[[0.]*4 for l in L]
            for cb in range(CB):
                acc=mma16(wfr[0][cb],g0[cb],acc); acc=mma16(wfr[1][cb],g1[cb],acc); acc=mma16(wfr[2][cb],g2[cb],acc)
            for l in L:
                l15=l&15; l4=l>>4
                if l4==1 and 1<=l15<=RC_STRIP and pc[l]<W:
                    gi[b,r,pc[l]]=[acc[(l15-1)&63][j]+acc[l][j]+acc[(32+l15+1)&63][j] for j in range(3)]
            g0=g1; g1=g2; r+=1
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
    print("fwd epi1 (LDS tile) err", np.abs(y-xb.detach().permute(0,2,3,1).numpy()).max())
    y,bits=fwdblur(imgn,wf,b0.detach().numpy(),B,H,W,CB,1)
    print("fwd+blur (row streaming) err", np.abs(y-xb.detach().permute(0,2,3,1).numpy()).max())
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

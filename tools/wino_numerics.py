#!/usr/bin/env python3
"""fp32 error of Winograd F(2x2,3x3) and F(4x4,3x3) against an fp64 direct convolution (128 input channels, numpy):
    python tools/wino_numerics.py      ->  direct fp32 2e-7, F(2x2) 2.5e-7, F(4x4) 3e-6 of the output scale"""
import numpy as np
rng=np.random.default_rng(0)
C=128; Co=32; H=W=16
x=rng.standard_normal((C,H+2,W+2)).astype(np.float32); x[:,0,:]=0; x[:,-1,:]=0; x[:,:,0]=0; x[:,:,-1]=0
w=(rng.standard_normal((Co,C,3,3))*0.02).astype(np.float32)
def direct(x,w,dt):
    x=x.astype(dt); w=w.astype(dt)
    out=np.zeros((w.shape[0],H,W),dt)
    for p in range(3):
        for q in range(3):
            out+=np.einsum('oc,chw->ohw',w[:,:,p,q],x[:,p:p+H,q:q+W]).astype(dt)
    return out
ref=direct(x,w,np.float64)
d32=direct(x,w,np.float32)
def wino(x,w,BT,G,AT,m):
    BT=BT.astype(np.float32); G=G.astype(np.float32); AT=AT.astype(np.float32)
    a=m+2
    U=np.einsum('ip,ocpq,jq->ijoc',G,w,G).astype(np.float32)
    out=np.zeros((w.shape[0],H,W),np.float32)
    for ty in range(H//m):
        for tx in range(W//m):
            d=x[:,ty*m:ty*m+a,tx*m:tx*m+a]
            V=np.einsum('ip,cpq,jq->ijc',BT,d,BT).astype(np.float32)
            M=np.einsum('ijoc,ijc->ijo',U,V).astype(np.float32)
            Y=np.einsum('ai,ijo,bj->oab',AT,M,AT).astype(np.float32)
            out[:,ty*m:(ty+1)*m,tx*m:(tx+1)*m]=Y
    return out
BT2=np.array([[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]],float); G2=np.array([[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]); AT2=np.array([[1,1,1,0],[0,1,-1,-1]],float)
BT4=np.array([[4,0,-5,0,1,0],[0,-4,-4,1,1,0],[0,4,-4,-1,1,0],[0,-2,-1,2,1,0],[0,2,-1,-2,1,0],[0,4,0,-5,0,1]],float)
G4=np.array([[1/4,0,0],[-1/6,-1/6,-1/6],[-1/6,1/6,-1/6],[1/24,1/12,1/6],[1/24,-1/12,1/6],[0,0,1]])
AT4=np.array([[1,1,1,1,1,0],[0,1,-1,2,-2,0],[0,1,1,4,4,0],[0,1,-1,8,-8,1]],float)
# better-conditioned points 0, 1, -1, 2, -1/2, inf (Cook-Toom construction): 0.6x the error, but the transform kernels
# lose their +-1 rows and the step gets 3 % slower (measured), so the kernels keep the textbook points
BT4b=np.array([[1,1.5,-2,-1.5,1,0],[0,-1,-2.5,-.5,1,0],[0,1,.5,-2.5,1,0],[0,-.5,-1,.5,1,0],[0,2,-1,-2,1,0],[0,1,1.5,-2,-1.5,1]])
G4b=np.array([[1,0,0],[-1/3,-1/3,-1/3],[1/3,-1/3,1/3],[1/15,2/15,4/15],[-16/15,8/15,-4/15],[0,0,1]])
AT4b=np.array([[1,1,1,1,1,0],[0,1,-1,2,-.5,0],[0,1,1,4,.25,0],[0,1,-1,8,-.125,1]],float)
s=np.abs(ref).max()
for name,o in (("direct fp32",d32),("F(2x2)",wino(x,w,BT2,G2,AT2,2)),("F(4x4) points 0,+-1,+-2 (used)",wino(x,w,BT4,G4,AT4,4)),
               ("F(4x4) points 0,1,-1,2,-1/2",wino(x,w,BT4b,G4b,AT4b,4))):
    print(name,"max err / scale", np.abs(o-ref).max()/s, "rms rel", np.sqrt(((o-ref)**2).mean())/np.sqrt((ref**2).mean()))

"""Numerical cost of a Winograd F(2x2, 3x3) form of the recogniser's 3x3 convolutions: emulates fp16 transformed weights
and fp16 transformed inputs with exact accumulation and compares with the direct form (fp16 weights, fp32 accumulation) on one layer.  CPU only."""
import numpy as np
rng=np.random.default_rng(0)
C,K,H,W=128,64,14,14
x=(rng.standard_normal((C,H+2,W+2))*0.5).astype(np.float16).astype(np.float32)   # padded input, fp16-representable
x[:,0,:]=x[:,-1,:]=0; x[:,:,0]=x[:,:,-1]=0
w=(rng.standard_normal((K,C,3,3))*(1/np.sqrt(9*C))).astype(np.float32)
w16=w.astype(np.float16).astype(np.float32)
# reference: float64 with the fp16-rounded weights and inputs (what the direct kernel computes up to fp32 accumulation error)
def direct(x,w):
    out=np.zeros((K,H,W),np.float64)
    for dy in range(3):
        for dx in range(3):
            out+=np.einsum('kc,chw->khw',w[:,:,dy,dx].astype(np.float64),x[:,dy:dy+H,dx:dx+W].astype(np.float64))
    return out
ref=direct(x,w16)
exact=direct(x,w)   # unrounded weights
# direct kernel emulation: fp32 accumulate
d32=np.zeros((K,H,W),np.float32)
for dy in range(3):
    for dx in range(3):
        d32+=np.einsum('kc,chw->khw',w16[:,:,dy,dx],x[:,dy:dy+H,dx:dx+W]).astype(np.float32)
# Winograd F(2x2,3x3)
G=np.array([[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]],np.float64)
Bt=np.array([[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]],np.float64)
At=np.array([[1,1,1,0],[0,1,-1,-1]],np.float64)
U=np.einsum('ai,kcij,bj->kcab',G,w.astype(np.float64),G)            # [K,C,4,4] from UNROUNDED fp32 weights
U16=U.astype(np.float16).astype(np.float64)
out=np.zeros((K,H,W),np.float64)
outv32=np.zeros((K,H,W),np.float64)
for ty in range(0,H,2):
    for tx in range(0,W,2):
        d=x[:,ty:ty+4,tx:tx+4].astype(np.float64)
        V=np.einsum('ai,cij,bj->cab',Bt,d,Bt)
        V16=V.astype(np.float16).astype(np.float64)                  # transformed input rounded to fp16 (what the MFMA consumes)
        M=np.einsum('kcab,cab->kab',U16,V16)
        out[:,ty:ty+2,tx:tx+2]=np.einsum('ia,kab,jb->kij',At,M,At)
def stats(name,a):
    e=a-exact
    print(name,'rel rms err vs exact-weights conv: %.3e'%(np.sqrt((e**2).mean())/np.sqrt((exact**2).mean())), ' max abs %.3e'%np.abs(e).max())
stats('direct fp16 weights, fp32 acc',d32.astype(np.float64))
stats('winograd fp16 U, fp16 V      ',out)
print('output rms',np.sqrt((exact**2).mean()),'fp16 ulp at that magnitude ~',np.sqrt((exact**2).mean())*2**-11)

"""Weighted minimax fit of erf(x / sqrt 2) ~ x * Q(x^2) on |x| <= xmax, pinned to 1 at xmax, evaluated in fp32 Horner form:
the coefficients of gelu_erf_poly2 (tango_amd/csrc/common.h).  Run: python tools/fit_gelu_poly.py"""
import numpy as np
from scipy.special import erf
def fit(xmax, deg, wend=30.0):
    n=6000
    th=(np.arange(n)+0.5)*np.pi/n
    x=(np.cos(th)+1)/2*xmax
    x=x[x>1e-6]
    x=np.concatenate([x,[xmax]])
    t=x*x
    y=erf(x/np.sqrt(2))
    y[-1]=1.0
    w=np.ones_like(x)
    V=np.vander(t/xmax**2,deg+1,increasing=True)*x[:,None]
    for it in range(80):
        ww=w.copy(); ww[-1]=wend
        c,*_=np.linalg.lstsq(V*ww[:,None],y*ww,rcond=None)
        e=np.abs(V@c-y)
        w[:-1]=(w*(e/e[:-1].max()+1e-3)**0.5)[:-1]
        w/=w[:-1].max()
    return c/(xmax**2)**np.arange(deg+1)
def ev32(c,x):
    x=x.astype(np.float32); xc=np.clip(x,-np.float32(XM),np.float32(XM)); t=xc*xc
    p=np.float32(c[-1])*np.ones_like(t)
    for k in range(len(c)-2,-1,-1): p=p*t+np.float32(c[k])
    e=p*xc
    h=np.float32(0.5)*x
    return h*e+h, e
for XM,deg in ((4.0,7),(4.0,8),(4.25,8),(4.25,9),(4.5,9),(4.0,9)):
    c=fit(XM,deg)
    x=np.linspace(-9,9,400001)
    g,e=ev32(c,x)
    ref=0.5*x*(1+erf(x/np.sqrt(2)))
    print(XM,deg,"erf err %.2e"%np.abs(e-erf(np.clip(x,-XM,XM)/np.sqrt(2))).max(),"gelu abs err %.2e"%np.abs(g-ref).max(), "e(XM)=%.8f"%e[-1])
    if (XM,deg)==(4.0,8) or (XM,deg)==(4.25,9): print("  coeffs:",", ".join("%.9ef"%v for v in c))

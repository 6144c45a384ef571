import mpmath as mp
mp.mp.dps = 60
def cheb_fit(f, a, b, deg):
    # Chebyshev interpolation at deg+1 nodes -> monomial coefficients (via solving Vandermonde in high precision)
    n = deg + 1
    xs = [ (a+b)/2 + (b-a)/2*mp.cos(mp.pi*(2*k+1)/(2*n)) for k in range(n)]
    A = mp.matrix(n, n); y = mp.matrix(n,1)
    for i,x in enumerate(xs):
        for j in range(n): A[i,j] = x**j
        y[i] = f(x)
    c = mp.lu_solve(A, y)
    return [c[i] for i in range(n)]
def maxerr(f, coef, a, b, rel=True, N=4001):
    worst = 0
    for i in range(N):
        x = a + (b-a)*i/(N-1)
        p = sum(c*x**j for j,c in enumerate(coef))
        fx = f(x)
        e = abs(p-fx)/abs(fx) if rel and fx != 0 else abs(p-fx)
        worst = max(worst, e)
    return worst
L = mp.log(2)/2
# exp(r) = 1 + r + r^2*P(r): fit P(r) = (exp(r)-1-r)/r^2
def P(r):
    if abs(r) < mp.mpf(10)**-15: return mp.mpf(1)/2 + r/6 + r*r/24
    return (mp.exp(r)-1-r)/r**2
for deg in (8,9,10):
    c = cheb_fit(P, -L*1.0001, L*1.0001, deg)
    cd = [mp.mpf(float(x)) for x in c]   # rounded to double
    full = [mp.mpf(1), mp.mpf(1)] + cd
    print("exp: P deg", deg, "total deg", deg+2, "rel err", mp.nstr(maxerr(mp.exp, full, -L, L),5))
    if deg == 9:
        print("EXP_P =", [float(x).hex() for x in c])
        print("EXP_P_dec =", [repr(float(x)) for x in c])
# log: m = (1+s)/(1-s); log m = 2 s + s^3 Q(s^2); Q(w) = (atanh(sqrt w)*2 - 2 sqrt w)/w^1.5
smax = (mp.sqrt(2)-1)/(mp.sqrt(2)+1)
wmax = smax**2 * 1.001
def Q(w):
    if abs(w) < mp.mpf(10)**-20: return mp.mpf(2)/3 + 2*w/5
    s = mp.sqrt(w)
    return (2*mp.atanh(s) - 2*s)/(s*w)
for deg in (5,6,7):
    c = cheb_fit(Q, 0, wmax, deg)
    cd = [mp.mpf(float(x)) for x in c]
    # error of full log relative
    worst = 0
    for i in range(1,4001):
        s = smax*i/4000
        w = s*s
        p = 2*s + s*w*sum(cc*w**j for j,cc in enumerate(cd))
        t = 2*mp.atanh(s)
        worst = max(worst, abs(p-t)/t)
    print("log: Q deg", deg, "rel err", mp.nstr(worst,5))
    if deg in (5,6,7):
        print("LOG_Q =", [float(x).hex() for x in c])
        print("LOG_Q_dec =", [repr(float(x)) for x in c])
print("smax", smax, "wmax", wmax)
ln2 = mp.log(2)
hi = float(ln2); 
# ln2_hi with trailing zeros for exact k*hi (k up to 1100: 11 bits)
import struct
def trunc(x, bits):
    i = struct.unpack('<q', struct.pack('<d', x))[0]
    i &= ~((1<<bits)-1)
    return struct.unpack('<d', struct.pack('<q', i))[0]
h = trunc(float(ln2), 12)   # 41 significant bits -> k*h exact for |k| < 2^12
l = float(ln2 - mp.mpf(h))
print("LN2_HI", h.hex(), repr(h), "LN2_LO", l.hex(), repr(l))
print("LOG2E", float(1/ln2).hex(), repr(float(1/ln2)))
print("LN2 full hi", float(ln2).hex(), "lo", float(ln2-mp.mpf(float(ln2))).hex())
print("SQRT2M1", float(mp.sqrt(2)-1).hex(), repr(float(mp.sqrt(2)-1)))

"""Derives the degree-6 polynomial of the filter's deterministic fp32 exp (oracle/rto_oracle.c orc_fexp,
csrc/rto_device_math.h fexp_f32*): weighted least squares with Remez-like reweighting of
(exp(r) - 1 - r) / r^2 on [-ln2/2, ln2/2], coefficients rounded to fp32, then the ulp error of the whole
fma-based evaluation measured against float64 exp on 4 M arguments."""
import numpy as np

f = np.float32


def fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def fexp(x, cs):
    x = np.maximum(x.astype(f), f(-88))
    t = fma(x, np.full_like(x, f(1.44269502162933349609375)), np.full_like(x, f(12582912.0)))
    kf = t - f(12582912.0)
    r = fma(kf, np.full_like(x, f(-0.693145751953125)), x)
    r = fma(kf, np.full_like(x, f(-1.42860676533018704e-06)), r)
    p = np.full_like(x, cs[0])
    for c in cs[1:]:
        p = fma(p, r, np.full_like(x, c))
    sc = ((kf.astype(np.int32) + 127) << 23).astype(np.uint32).view(np.float32)
    return p * sc


def main():
    rs = np.random.RandomState(0)
    xs = np.concatenate([rs.uniform(-87.3, 0, 3000000), rs.uniform(-2, 0, 1000000)]).astype(f)
    ref = np.exp(xs.astype(np.float64))
    ulp = np.spacing(ref.astype(f)).astype(np.float64)
    a = 0.34658 * 1.001
    r = np.cos(np.pi * (np.arange(4000) + 0.5) / 4000) * a
    q = np.where(np.abs(r) > 1e-8, (np.expm1(r) - r) / np.where(r == 0, 1, r) ** 2, 0.5)
    w = r ** 2 / np.exp(r)
    c = np.polyfit(r, q, 4, w=w)
    for _ in range(30):
        err = (np.polyval(c, r) - q) * w
        w2 = w * (1 + 3 * np.abs(err) / np.abs(err).max())
        c = np.polyfit(r, q, 4, w=w2)
        w = w2 / w2.max() * ((r ** 2 / np.exp(r)).max())
    cs = [f(v) for v in c] + [f(1.0), f(1.0)]
    err = np.abs(fexp(xs, cs).astype(np.float64) - ref) / ulp
    print("max ulp %.3f mean %.3f" % (err.max(), err.mean()))
    print([float(v).hex() for v in cs])


if __name__ == "__main__":
    main()

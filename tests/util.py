import numpy as np

# stated fp32 tolerances: the GPU reduces in a different order than the sequential oracle
ACT_REL_L2 = 1e-5
ACT_MAX_ABS = 1e-4
GRAD_REL_L2 = 1e-4
LOSS_ABS = 1e-4


def nhwc(a):
    return np.ascontiguousarray(np.transpose(a, (0, 2, 3, 1)))


def nchw(a):
    return np.ascontiguousarray(np.transpose(a, (0, 3, 1, 2)))


def rel_l2(a, b):
    a = np.asarray(a, np.float64).ravel()
    b = np.asarray(b, np.float64).ravel()
    return float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-30))


def max_abs_rel(a, b):
    a = np.asarray(a, np.float64).ravel()
    b = np.asarray(b, np.float64).ravel()
    return float(np.max(np.abs(a - b)) / (np.max(np.abs(b)) + 1e-30))


def check_act(got, ref, what, rel=ACT_REL_L2, mx=ACT_MAX_ABS):
    r, m = rel_l2(got, ref), max_abs_rel(got, ref)
    assert r <= rel and m <= mx, "%s: rel-L2 %.3e (tol %.1e), max-abs/max|ref| %.3e (tol %.1e)" % (what, r, rel, m, mx)


def check_grad(got, ref, what, rel=GRAD_REL_L2):
    r = rel_l2(got, ref)
    assert r <= rel, "%s: rel-L2 %.3e (tol %.1e)" % (what, r, rel)


def rand(shape, seed, scale=1.0):
    import synth
    n = int(np.prod(shape))
    return (synth.normal(seed, n, 1.0) * scale).reshape(shape).astype(np.float32)

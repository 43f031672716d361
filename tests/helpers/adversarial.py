"""Adversarial inputs for the candidate filter (K4h) -- TEST INFRASTRUCTURE.

The filter keeps a (row, query) pair unless its f16 / bf16 matrix-core score is below L_q - E_q(R_t); its answer is exact
only while E_q really bounds the approximation's error.  Gaussian data never gets near that bound (rounding errors of
hundreds of elements cancel).  The rows built here do:

  * every element of a row and of a query sits a hair off an f16 (bf16) ROUNDING MIDPOINT, on the side that makes the
    conversion err by (almost) half an ulp in a chosen direction;
  * "A" rows -- among them the true neighbours -- are rounded so that every product LOSES (aligned elements shrink,
    opposed elements grow): their approximate score is as far BELOW the exact one as the format allows;
  * "B" rows are rounded the other way (every product gains): the sample's witnesses, from which L_q is made, are as far
    ABOVE their exact scores as the format allows;
  * A and B rows have the same number of sign flips against the query, so thousands of rows lie within a fraction of one
    margin of the k-th best exact score, and further levels follow one margin apart.
"""
import numpy as np
import torch


def _grid(bf16):
    """(mantissa step of the 16-bit format relative to the binade, number of mantissa codes used)"""
    return (2.0 ** -7, 8) if bf16 else (2.0 ** -10, 8)


def make_elements(gen, shape, base, bf16, direction, boost=None):
    """Magnitudes base * (1 + m * step + off): off just below (direction < 0: rounds toward zero) or just above
    (direction > 0: rounds away from zero) the midpoint between two codes; direction == 0: exactly on the grid.
    boost: the fraction of elements one code larger (a slightly better exact score).  Exact in f32 (12 significant bits)."""
    step, codes = _grid(bf16)
    dev = gen.device
    m = torch.randint(0, codes, shape, generator=gen, device=dev).float()
    if boost is not None:
        m = m + (torch.rand(shape, generator=gen, device=dev) < boost).float()
    d = direction if torch.is_tensor(direction) else torch.full(shape, float(direction), device=dev)
    lo, hi = 0.5 * step * (1 - 2.0 ** -6), 0.5 * step * (1 + 2.0 ** -6)
    off = torch.where(d < 0, lo, 0.0) + torch.where(d > 0, hi, 0.0)
    return base * (1.0 + m * step + off)


def build(seed, dim, n_queries, k, bf16_rows, a_per_query=12, level0=300, levels=18, per_level=90, base_exp=None,
          dump_rows=4096, exact_flips=True, device=None):
    """-> rows [n][dim] f32, queries [nq][dim] f32, (row -> query it was built for) [n], (is an A row) [n]   (numpy, host).
    The A rows all lie in the first dump_rows rows (the region the experiments build dumps its gate's view of).
    Generated with torch on `device` (the GPU when there is one: the sweep makes a thousand of these)."""
    if device is None:
        device = torch.device("cuda", 0) if torch.cuda.is_available() else torch.device("cpu")
    gen = torch.Generator(device=device)
    gen.manual_seed(int(seed))
    if base_exp is None:
        base_exp = -int(np.ceil(np.log2(np.sqrt(dim))))          # row norms a little below 1
    base = 2.0 ** base_exp
    sq = torch.randint(0, 2, (n_queries, dim), generator=gen, device=device).float() * 2 - 1
    # queries: f32 rows -> the f16 pipe rounds the query (toward zero here: every aligned product loses);
    # bf16 rows on the bf16 matrix cores -> the query is rounded to bf16, same construction on that grid
    Q = sq * make_elements(gen, (n_queries, dim), base, bf16_rows, -1)
    f0 = dim // 8
    # one family per (query, level): count, sign flips against the query, zeroed elements (half a flip), rounding direction
    fam = [(q, a_per_query, f0, 0, -1) for q in range(n_queries)]                       # A: the true neighbours' kind
    n_a = a_per_query * n_queries
    for q in range(n_queries):
        fam.append((q, level0, f0, 0, +1))                                              # B, same exact level as A
        for lv in range(1, levels + 1):                                                 # further levels about one margin apart
            fam.append((q, per_level, f0 + lv // 2, lv % 2, +1 if lv % 3 else -1))
    fam = np.asarray(fam)
    owner = np.repeat(fam[:, 0], fam[:, 1])
    n = owner.shape[0]
    is_a = np.arange(n) < n_a
    t_owner = torch.from_numpy(owner).to(device)
    flips = torch.from_numpy(np.repeat(fam[:, 2], fam[:, 1]).astype(np.float32)).to(device)[:, None]
    zeros = torch.from_numpy(np.repeat(fam[:, 3], fam[:, 1]).astype(np.float32)).to(device)[:, None]
    dirn = torch.from_numpy(np.repeat(fam[:, 4], fam[:, 1]).astype(np.float32)).to(device)[:, None]
    boost = torch.from_numpy(np.where(is_a, np.float32(0.3), np.float32(0))).to(device)[:, None]
    X = torch.empty((n, dim), device=device)
    for lo in range(0, n, 32768):
        hi = min(n, lo + 32768)
        sqr = sq[t_owner[lo:hi]]
        u = torch.rand((hi - lo, dim), generator=gen, device=device)
        # exact: a random permutation per row (its first `flips` places flip); fast: the number of flips is binomial around it
        rank = u.argsort(1).argsort(1).float() if exact_flips else u * dim
        t = torch.where(rank < flips[lo:hi], -sqr, torch.where(rank < flips[lo:hi] + zeros[lo:hi], torch.zeros_like(sqr), sqr))
        # the direction applies to ALIGNED elements; opposed elements get the opposite rounding so that the product moves the same way
        al = torch.where(t * sqr > 0, dirn[lo:hi], -dirn[lo:hi]).expand(hi - lo, dim)
        # bf16 rows are stored rounded: they sit exactly on the grid (the row side has no conversion error on the bf16 pipe)
        X[lo:hi] = t * make_elements(gen, (hi - lo, dim), base, bf16_rows, 0 if bf16_rows else al, boost=boost[lo:hi])
    assert n_a <= dump_rows <= n
    # A rows at random places inside the dump region, everything else shuffled over the rest
    rng = np.random.default_rng(int(seed))
    rest = rng.permutation(np.arange(n_a, n))
    head = rng.permutation(np.concatenate([np.arange(n_a), rest[:dump_rows - n_a]]))
    order = np.concatenate([head, rest[dump_rows - n_a:]])
    X = X[torch.from_numpy(order).to(device)]
    return X.cpu().numpy(), Q.cpu().numpy(), owner[order], is_a[order]


def bf16_round(x):
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)

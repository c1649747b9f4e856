"""TEST INFRASTRUCTURE ONLY (see oracle/README or DESIGN.md §4): exact Toom-Cook construction of the 1-D minimal
filtering transforms F(m, r) the Winograd kernels hard-code (csrc/conv_wino7.hip: struct WT; csrc/conv_wino.hip).

For interpolation points p_0 .. p_{n-2} and infinity (n = m + r - 1):
    y = AT [ (G g) o (BT d) ],   y_i = sum_k d_{i+k} g_k   (correlation, m outputs, r taps, n inputs)
with  AT[i][j] = p_j^i (last column: [i == m-1]),  G[j][k] = p_j^k / N_j (last row: [k == r-1]),
      BT = (V^-1)^T with rows scaled by N_j,  V = evaluation matrix of degree n-1 polynomials at the points,
      N_j = prod_{l != j} (p_j - p_l).
Everything in exact rational arithmetic (fractions)."""
from fractions import Fraction as Fr


def toom_cook(m, r, points):
    n = m + r - 1
    assert len(points) == n - 1
    pts = [Fr(p) for p in points]
    V = [[p ** k for k in range(n)] for p in pts] + [[Fr(0)] * (n - 1) + [Fr(1)]]
    M = [row[:] + [Fr(int(i == j)) for j in range(n)] for i, row in enumerate(V)]
    for c in range(n):
        piv = next(i for i in range(c, n) if M[i][c] != 0)
        M[c], M[piv] = M[piv], M[c]
        pv = M[c][c]
        M[c] = [x / pv for x in M[c]]
        for i in range(n):
            if i != c and M[i][c] != 0:
                f = M[i][c]
                M[i] = [a - f * b for a, b in zip(M[i], M[c])]
    Vinv = [row[n:] for row in M]
    BT = [[Vinv[j][i] for j in range(n)] for i in range(n)]
    G = [[p ** k for k in range(r)] for p in pts] + [[Fr(0)] * (r - 1) + [Fr(1)]]
    AT = [[p ** i for p in pts] + [Fr(int(i == m - 1))] for i in range(m)]
    for j, p in enumerate(pts):
        N = Fr(1)
        for l, q in enumerate(pts):
            if l != j:
                N *= p - q
        BT[j] = [x * N for x in BT[j]]
        G[j] = [x / N for x in G[j]]
    return AT, G, BT


# the point sets of the kernels
POINTS_F4_7 = [0, 1, -1, 2, -2, Fr(1, 2), Fr(-1, 2), Fr(3, 2), Fr(-3, 2)]
POINTS_F6_7 = POINTS_F4_7 + [Fr(2, 3), Fr(-2, 3)]
POINTS_F2_3 = [0, 1, -1]
POINTS_F4_3 = [0, Fr(3, 4), Fr(-3, 4), Fr(3, 2), Fr(-3, 2)]   # F(4x4,3x3), csrc/conv_wino4.hip

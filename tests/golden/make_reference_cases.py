"""Extracts the NUMERIC test inputs (μ, d, C / Σ literals) of the reference's statistical tests
(/root/reference/test/sample-correctness_tests.jl:27-118) into tests/golden/reference_mvn_cases.json,
reference_mixture_case.json and reference_tail_cases.json.
Only data is kept — each case becomes {name, mu, L} with x = μ + L z, z ~ N(0, I) (the reference's
multivariate_normal(μ, L), test/utilities.jl:64-67); no source text is copied.  Run in the build container
(the reference is not present on the GPU box)."""
import json
import os
import re

import numpy as np

SRC = "/root/reference/test/sample-correctness_tests.jl"
HERE = os.path.dirname(os.path.abspath(__file__))


def parse_matrix(txt):
    rows = [r.strip() for r in txt.replace("\n", " ").split(";")]
    return np.array([[float(x) for x in r.split()] for r in rows if r])


def main():
    src = open(SRC).read()
    cases = []
    # --- ill-conditioned: μ = [...], d = [...], C = [...] possibly followed by ' (transpose) -------------
    sec = src[src.index('@testset "ill-conditioned multivariate normal"'):src.index('@testset "NUTS tests with specific normal distributions"')]
    mus = re.findall(r"μ = \[([^\]]+)\]", sec)
    ds = re.findall(r"\bd = \[([^\]]+)\]", sec)
    Cs = re.findall(r"C = \[([^\]]+)\]('?)", sec)
    mu_for = [mus[0], mus[0], mus[1]]          # the second case reuses the first μ
    for i in range(3):
        mu = np.array([float(x) for x in mu_for[i].split(",")])
        d = np.array([float(x) for x in ds[i].split(",")])
        C = parse_matrix(Cs[i][0])
        if Cs[i][1] == "'":
            C = C.T
        cases.append(dict(name=f"ill-conditioned {i + 1}", mu=mu.tolist(), L=(np.diag(d) @ C).tolist(), metric="Symmetric"))
    # --- specific normals -------------------------------------------------------------------------------
    cases.append(dict(name="univariate huge variance", mu=[0.0], L=[[5e8]], metric="Diagonal"))
    cases.append(dict(name="univariate huge variance, offset", mu=[1.0], L=[[5e8]], metric="Diagonal"))
    cases.append(dict(name="univariate tiny variance, offset", mu=[1.0], L=[[5e-8]], metric="Diagonal"))
    cases.append(dict(name="mildly scaled diagonal", mu=[1.0, 2.0, 3.0], L=np.diag([1.0, 2.0, 3.0]).tolist(), metric="Diagonal"))
    sec = src[src.index('@testset "NUTS tests with specific normal distributions"'):src.index('@testset "NUTS tests with mixtures"')]
    for m in re.finditer(r"multivariate_normal\(\s*\[([^\]]+)\],\s*cholesky\(\[([^\]]+)\]\)\.L\)", sec):
        mu = np.array([float(x) for x in m.group(1).split(",")])
        S = parse_matrix(m.group(2))
        cases.append(dict(name=f"kept {len(mu)} dim", mu=mu.tolist(), L=np.linalg.cholesky(S).tolist(), metric="Diagonal"))
    with open(os.path.join(HERE, "reference_mvn_cases.json"), "w") as fh:
        json.dump(cases, fh, indent=0)
    # --- mixture of two normals (:89-98): α, the second component's mean and L = D2 * C2, and the alert levels of the call ----
    sec = src[src.index('@testset "NUTS tests with mixtures"'):src.index('@testset "NUTS tests with heavier tails and skewness"')]
    C2 = parse_matrix(re.search(r"C2 = \[([^\]]+)\]", sec).group(1))
    d2 = float(re.search(r"D2 = I \* ([0-9.]+)", sec).group(1))
    alpha = float(re.search(r"mix\(([0-9.]+),", sec).group(1))
    tau_alert = float(re.search(r"τ_alert = ([0-9.]+)", sec).group(1))
    p_alert = float(re.search(r"p_alert = ([0-9.]+)", sec).group(1))
    mix = dict(name="mixture of two normals", alpha=alpha, mu1=[0.0] * 3, L1=np.eye(3).tolist(), mu2=[1.0] * 3, L2=(d2 * C2).tolist(),
               tau_alert=tau_alert, p_alert=p_alert, N=1000)
    with open(os.path.join(HERE, "reference_mixture_case.json"), "w") as fh:
        json.dump(mix, fh, indent=0)
    print(mix["name"], "alpha", alpha, "tau_alert", tau_alert, "p_alert", p_alert)
    # --- heavier tails and skewness (:100-118): the numbers of the three calls — K, the elongation exponent, the shift, the mixture
    # weight, N and each call's alert / fail levels.  (What elongate / shift / funnel MEAN is LogDensityTestSuite's, which is not
    # under /root/reference: the tests that replay these cases write their own definitions down.)
    sec = src[src.index('@testset "NUTS tests with heavier tails and skewness"'):]
    K = int(re.search(r"K = (\d+)", sec).group(1))
    calls = re.findall(r'NUTS_tests\(RNG, ℓ, "([^"]+)",\s*(\d+);([^)]*)\)', sec)
    tails = []
    for title, N, kw in calls:
        levels = {k: float(v) for k, v in re.findall(r"(\S+) = ([0-9.e-]+)", kw)}
        tails.append(dict(name=title, N=int(N), levels=levels))
    ks = [float(x) for x in re.findall(r"elongate\(([0-9.]+)\)", sec)]
    assert len(set(ks)) == 1
    out = dict(K=K, elongate=ks[0], shift=[1.0] * K if "shift(ones(K))" in sec else None,
               funnel_mix_alpha=float(re.search(r"mix\(([0-9.]+), funnel\(\)", sec).group(1)), calls=tails)
    with open(os.path.join(HERE, "reference_tail_cases.json"), "w") as fh:
        json.dump(out, fh, indent=0, ensure_ascii=False)
    print(out)
    for c in cases:
        print(c["name"], len(c["mu"]))


if __name__ == "__main__":
    main()

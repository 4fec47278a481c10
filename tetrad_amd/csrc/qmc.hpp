// qmc.hpp -- clean-room weighted Quartet MaxCut supertree (host code; SURVEY.md section 8 row f3, second half).
// Part of the single translation unit tetrad_hip.hip (included inside its anonymous namespace).
//
// Takes the place of the step right after the hot path: `run_qmc` (tetrad/src/run_inference.py:146-166)
// shells out to the prebuilt `bin/max-cut-tree qrtt=<file> weights=on|off otre=<file>` -- a binary with no
// source in the reference, which is never run or read here.  This is an implementation written from the
// PUBLISHED description of the method (Snir & Rao, "Quartets MaxCut: a divide and conquer quartets
// algorithm", IEEE/ACM TCBB 2010; Avni, Cohen & Snir, "Weighted quartets phylogenetics", Syst. Biol. 2015):
//   * the quartets ab|cd over the current taxon set define a graph with "good" edges (a-c, a-d, b-c, b-d:
//     the cut should separate them) and "bad" edges (a-b, c-d: it should not);
//   * a bipartition that maximises good-cut weight against bad-cut weight is sought (here: maximise
//     good - alpha * bad by multi-start local search, alpha raised Dinkelbach-style to the ratio of the best
//     cut found -- the published method scans alpha on an embedding of the graph on a sphere);
//   * quartets inside one side are passed down, quartets with three taxa on one side keep those three plus
//     an artificial taxon that stands for the other side, the rest (satisfied or violated) are dropped;
//   * the two sub-trees are joined at their artificial taxa.
// PARITY IS UNPINNED BY CONSTRUCTION: there is no reference source, the reference's tests hold no tree for
// any quartet set, and its binary may not be executed.  What is tested is what any correct implementation
// must do: recover the generating tree from its own (complete or sampled, weighted or not) quartet set.
#pragma once

struct QmcQuartet {
    int32_t t[4];       // t0,t1 | t2,t3
    double w;
};

struct QmcForest {
    // unrooted trees in one arena: label >= 0 = taxon (artificial ones are >= ntaxa), -1 = internal node
    std::vector<int32_t> label;
    std::vector<std::vector<int32_t>> adj;
    std::unordered_map<int32_t, int32_t> leaf_of;       // taxon label -> node
    int32_t add(int32_t lab)
    {
        label.push_back(lab);
        adj.emplace_back();
        const int32_t id = (int32_t)label.size() - 1;
        if (lab >= 0) leaf_of[lab] = id;
        return id;
    }
    void link(int32_t a, int32_t b)
    {
        adj[a].push_back(b);
        adj[b].push_back(a);
    }
    void unlink(int32_t a, int32_t b)
    {
        auto rm = [](std::vector<int32_t> &v, int32_t x) {
            for (size_t i = 0; i < v.size(); ++i)
                if (v[i] == x) {
                    v[i] = v.back();
                    v.pop_back();
                    return;
                }
        };
        rm(adj[a], b);
        rm(adj[b], a);
    }
};

struct QmcRng {
    uint64_t s;
    uint64_t next()
    {
        s += 0x9E3779B97F4A7C15ull;
        uint64_t z = s;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        return z ^ (z >> 31);
    }
    double unit() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
};

// value of a bipartition: weight of cut good edges and of cut bad edges
inline void qmc_cut_weights(const std::vector<double> &G, const std::vector<double> &B, const std::vector<uint8_t> &side,
                            int n, double &good, double &bad)
{
    good = bad = 0.0;
    for (int u = 0; u < n; ++u)
        for (int v = u + 1; v < n; ++v)
            if (side[u] != side[v]) {
                good += G[(size_t)u * n + v];
                bad += B[(size_t)u * n + v];
            }
}

// local search for max  sum_{u,v separated} (G - alpha*B)[u][v]  with both sides >= 2 taxa
inline double qmc_local_search(const std::vector<double> &G, const std::vector<double> &B, double alpha, int n,
                               std::vector<uint8_t> &side)
{
    // gain[v] = change of the objective when v changes side = sum_same W - sum_other W
    std::vector<double> gain(n, 0.0);
    int cnt[2] = {0, 0};
    for (int v = 0; v < n; ++v) cnt[side[v]]++;
    for (int v = 0; v < n; ++v) {
        double g = 0.0;
        for (int u = 0; u < n; ++u) {
            if (u == v) continue;
            const double w = G[(size_t)v * n + u] - alpha * B[(size_t)v * n + u];
            g += (side[u] == side[v]) ? w : -w;
        }
        gain[v] = g;
    }
    for (int pass = 0; pass < 50 * n; ++pass) {
        int best = -1;
        double bg = 1e-12;
        for (int v = 0; v < n; ++v)
            if (gain[v] > bg && cnt[side[v]] > 2) {
                bg = gain[v];
                best = v;
            }
        if (best < 0) break;
        const int v = best;
        cnt[side[v]]--;
        side[v] ^= 1;
        cnt[side[v]]++;
        gain[v] = -gain[v];
        for (int u = 0; u < n; ++u) {
            if (u == v) continue;
            const double w = G[(size_t)v * n + u] - alpha * B[(size_t)v * n + u];
            // u and v are now on the same side iff side[u] == side[v]
            gain[u] += (side[u] == side[v]) ? 2.0 * w : -2.0 * w;
        }
    }
    double good, bad;
    qmc_cut_weights(G, B, side, n, good, bad);
    return good - alpha * bad;
}

// best bipartition of n >= 4 vertices; false when the quartets carry no signal at all
inline bool qmc_best_cut(const std::vector<QmcQuartet> &qs, const std::vector<int32_t> &index_of_label_dense,
                         int n, QmcRng &rng, std::vector<uint8_t> &best_side)
{
    std::vector<double> G((size_t)n * n, 0.0), B((size_t)n * n, 0.0);
    auto addw = [&](std::vector<double> &M, int u, int v, double w) {
        M[(size_t)u * n + v] += w;
        M[(size_t)v * n + u] += w;
    };
    double total = 0.0;
    for (const auto &q : qs) {
        const int a = index_of_label_dense[q.t[0]], b = index_of_label_dense[q.t[1]], c = index_of_label_dense[q.t[2]],
                  d = index_of_label_dense[q.t[3]];
        addw(B, a, b, q.w);
        addw(B, c, d, q.w);
        addw(G, a, c, q.w);
        addw(G, a, d, q.w);
        addw(G, b, c, q.w);
        addw(G, b, d, q.w);
        total += q.w;
    }
    if (!(total > 0.0)) return false;
    if (n == 4) {                                                   // the three 2|2 splits, exhaustively
        double best = -1.0;
        for (int k = 1; k < 4; ++k) {
            std::vector<uint8_t> s4(4, 1);
            s4[0] = 0;
            s4[k] = 0;
            double good, bad;
            qmc_cut_weights(G, B, s4, 4, good, bad);
            const double ratio = good / (bad + 1e-9 * total);
            if (good > 0.0 && ratio > best) {
                best = ratio;
                best_side = s4;
            }
        }
        return best >= 0.0;
    }
    double best_ratio = -1.0, alpha = 1.0;
    std::vector<uint8_t> side(n);
    const int starts = n <= 8 ? 24 : 12;
    for (int round = 0; round < 6; ++round) {
        bool improved = false;
        for (int s = 0; s < starts + 1; ++s) {
            if (s == 0 && best_ratio >= 0.0) {
                side = best_side;                                   // refine the incumbent under the new alpha
            } else {
                int c1 = 0;
                for (int v = 0; v < n; ++v) c1 += (side[v] = (uint8_t)(rng.next() & 1));
                if (c1 < 2 || n - c1 < 2) {                          // force a valid start
                    for (int v = 0; v < n; ++v) side[v] = (uint8_t)(v & 1);
                }
            }
            qmc_local_search(G, B, alpha, n, side);
            double good, bad;
            qmc_cut_weights(G, B, side, n, good, bad);
            if (good <= 0.0) continue;
            const double ratio = good / (bad + 1e-9 * total);
            if (ratio > best_ratio * (1.0 + 1e-12)) {
                best_ratio = ratio;
                best_side = side;
                improved = true;
            }
        }
        if (best_ratio < 0.0) return false;
        if (!improved && round > 0) break;
        double good, bad;
        qmc_cut_weights(G, B, best_side, n, good, bad);
        if (bad <= 1e-12 * total) break;                            // nothing is violated: cannot do better
        alpha = good / bad;                                         // Dinkelbach step for max good / bad
    }
    return best_ratio >= 0.0;
}

inline int32_t qmc_star(QmcForest &F, const std::vector<int32_t> &taxa)
{
    const int32_t c = F.add(-1);
    for (int32_t t : taxa) F.link(c, F.add(t));
    return c;
}

// builds the tree of `taxa` from `qs` inside F; returns one of its nodes
inline int32_t qmc_build(QmcForest &F, const std::vector<int32_t> &taxa, std::vector<QmcQuartet> &qs, int32_t &next_label,
                         QmcRng &rng, std::vector<int32_t> &dense, int depth)
{
    const int n = (int)taxa.size();
    if (n <= 3 || qs.empty() || depth > 4096) return qmc_star(F, taxa);
    for (int i = 0; i < n; ++i) dense[taxa[i]] = i;
    std::vector<uint8_t> side;
    if (!qmc_best_cut(qs, dense, n, rng, side)) return qmc_star(F, taxa);
    const int32_t artA = next_label++, artB = next_label++;          // artA stands for side 0, artB for side 1
    if ((size_t)next_label > dense.size()) dense.resize((size_t)next_label + 64, 0);
    std::vector<int32_t> A, Bt;
    for (int i = 0; i < n; ++i) (side[i] ? Bt : A).push_back(taxa[i]);
    A.push_back(artB);
    Bt.push_back(artA);
    std::vector<QmcQuartet> qa, qb;
    for (const auto &q : qs) {
        int s[4], ones = 0;
        for (int k = 0; k < 4; ++k) ones += (s[k] = side[dense[q.t[k]]]);
        if (ones == 0) qa.push_back(q);
        else if (ones == 4) qb.push_back(q);
        else if (ones == 1) {                                        // three on side 0: the odd one becomes artB
            QmcQuartet r = q;
            for (int k = 0; k < 4; ++k)
                if (s[k]) r.t[k] = artB;
            qa.push_back(r);
        } else if (ones == 3) {
            QmcQuartet r = q;
            for (int k = 0; k < 4; ++k)
                if (!s[k]) r.t[k] = artA;
            qb.push_back(r);
        }
    }
    std::vector<QmcQuartet>().swap(qs);                              // release before recursing
    qmc_build(F, A, qa, next_label, rng, dense, depth + 1);
    qmc_build(F, Bt, qb, next_label, rng, dense, depth + 1);
    // join the two trees at their artificial leaves
    const int32_t la = F.leaf_of[artB], lb = F.leaf_of[artA];
    const int32_t pa = F.adj[la][0], pb = F.adj[lb][0];
    F.unlink(la, pa);
    F.unlink(lb, pb);
    F.link(pa, pb);
    return pa;
}

// newick of the component of `node` entered from `from` (unary internal nodes are passed through)
inline void qmc_newick(const QmcForest &F, int32_t node, int32_t from, std::string &out)
{
    std::vector<int32_t> kids;
    for (int32_t v : F.adj[node])
        if (v != from) kids.push_back(v);
    if (F.label[node] >= 0) {
        out += std::to_string(F.label[node]);
        return;
    }
    if (kids.size() == 1) {
        qmc_newick(F, kids[0], node, out);
        return;
    }
    // deterministic order: by smallest taxon below -- cheap proxy: recurse then sort the pieces
    std::vector<std::string> parts(kids.size());
    for (size_t i = 0; i < kids.size(); ++i) qmc_newick(F, kids[i], node, parts[i]);
    out += '(';
    for (size_t i = 0; i < parts.size(); ++i) {
        if (i) out += ',';
        out += parts[i];
    }
    out += ')';
}

// splits u32[n,4] = a,b|c,d; weights f64[n] or null (all 1).  Returns the newick text (with ';').
inline int qmc_tree(const uint32_t *splits, const double *weights, int64_t n, int64_t ntaxa, uint64_t seed,
                    std::string &newick)
{
    if (ntaxa < 1) return TQ_ERR_INVALID_ARG;
    std::vector<QmcQuartet> qs;
    qs.reserve((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        QmcQuartet q;
        bool ok = true;
        for (int k = 0; k < 4; ++k) {
            if (splits[i * 4 + k] >= (uint64_t)ntaxa) return TQ_ERR_INVALID_ARG;
            q.t[k] = (int32_t)splits[i * 4 + k];
        }
        for (int k = 0; k < 4 && ok; ++k)
            for (int j = k + 1; j < 4; ++j)
                if (q.t[k] == q.t[j]) ok = false;
        q.w = weights ? weights[i] : 1.0;
        if (ok && q.w > 0.0 && std::isfinite(q.w)) qs.push_back(q);
    }
    std::vector<int32_t> taxa((size_t)ntaxa);
    for (int64_t i = 0; i < ntaxa; ++i) taxa[(size_t)i] = (int32_t)i;
    QmcForest F;
    QmcRng rng{seed ^ 0xA5A5A5A5DEADBEEFull};
    int32_t next_label = (int32_t)ntaxa;
    std::vector<int32_t> dense((size_t)ntaxa * 3 + 64, 0);
    const int32_t node = qmc_build(F, taxa, qs, next_label, rng, dense, 0);
    newick.clear();
    if (ntaxa == 1) {
        newick = "0;";
        return TQ_OK;
    }
    // root: at the last join edge (node -- its neighbour across the join) when there is one, else at the star centre
    int32_t root = node;
    if (F.label[root] >= 0) root = F.adj[root].empty() ? root : F.adj[root][0];
    qmc_newick(F, root, -1, newick);
    newick += ';';
    return TQ_OK;
}

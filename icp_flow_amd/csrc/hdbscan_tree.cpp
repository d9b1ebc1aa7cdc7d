// hdbscan_tree.cpp -- the sequential remainder of HDBSCAN on the n - 1 edges of the spanning tree (HOST code).
//
// What hdbscan.HDBSCAN does after its spanning tree (the reference calls it at utils_cluster.py:12-17; third-party
// `hdbscan` 0.8.29, environment.yml:57; scikit-learn's port of the same steps: sklearn/cluster/_hdbscan/hdbscan.py
// _process_mst, _linkage.pyx make_single_linkage, _tree.pyx _condense_tree / _compute_stability / _get_clusters /
// _do_labelling), restated from the published algorithm (Campello, Moulavi, Sander 2013; McInnes, Healy 2017)
// with the defaults the reference uses: cluster_selection_method "eom", allow_single_cluster False,
// cluster_selection_epsilon 0, no max_cluster_size:
//   1. edges directed away from point 0 (Prim's emission order: endpoint already in the tree, new endpoint),
//      sorted by (weight, first endpoint, second endpoint) -- a fixed order for ties;
//   2. single-linkage dendrogram by union-find (merge i creates node n + i);
//   3. condensed tree, breadth first from the root: a split counts when both sides hold >= min_cluster_size
//      points, otherwise the small side's points fall out of the parent at lambda = 1 / distance;
//   4. stability = sum over the rows of a cluster of (lambda - lambda_birth) * size, accumulated in row order;
//   5. excess of mass, children before parents: a cluster survives unless its children's stabilities sum higher;
//   6. a point takes the label of its nearest selected ancestor (labels = rank of the selected cluster ids),
//      -1 if there is none.
// All of it is O(n log n) pointer chasing on ~10^5 edges: a few milliseconds on one host core, not device work.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <numeric>
#include <vector>

#include "../../include/icpflow_hip.h"

namespace {

struct Row {   // condensed tree
    int parent, child;
    double lambda;
    int size;
};

struct Edge {
    double w;
    int cur, nxt;
};

// The work arrays (~130 bytes per point) are kept per host thread between calls: allocating and first touching 15 MB
// per frame pair costs about a millisecond of page faults, a quarter of what the rest of the function takes.
struct Scratch {
    std::vector<int> start, adj, fill, pred, queue, par, top, left, right, count, first, leafOrder, relabel, bfs, sub;
    std::vector<Edge> edges, tmp;
    std::vector<double> dist;
    std::vector<Row> rows;
};

}  // namespace

extern "C" int icpflow_hdbscan_labels(const int32_t *h_edge_a, const int32_t *h_edge_b, const double *h_edge_w,
                                      int n_points, int min_cluster_size, int32_t *h_labels)
{
    const int n = n_points;
    if (!h_labels || n <= 0 || min_cluster_size < 2) return ICPFLOW_E_ARG;
    if (n == 1) {
        h_labels[0] = -1;
        return 0;
    }
    if (!h_edge_a || !h_edge_b || !h_edge_w) return ICPFLOW_E_ARG;
    const int m = n - 1;
    for (int e = 0; e < m; ++e)
        if (h_edge_a[e] < 0 || h_edge_a[e] >= n || h_edge_b[e] < 0 || h_edge_b[e] >= n || h_edge_a[e] == h_edge_b[e])
            return ICPFLOW_E_ARG;

    // 1. orientation away from point 0
    static thread_local Scratch S;
    std::vector<int> &start = S.start, &adj = S.adj;
    start.assign(n + 1, 0);
    adj.resize(2 * (size_t)m);
    for (int e = 0; e < m; ++e) {
        ++start[h_edge_a[e] + 1];
        ++start[h_edge_b[e] + 1];
    }
    for (int i = 0; i < n; ++i) start[i + 1] += start[i];
    {
        std::vector<int> &fill = S.fill;
        fill.assign(start.begin(), start.end() - 1);
        for (int e = 0; e < m; ++e) {
            adj[fill[h_edge_a[e]]++] = h_edge_b[e];
            adj[fill[h_edge_b[e]]++] = h_edge_a[e];
        }
    }
    std::vector<int> &pred = S.pred, &queue = S.queue;
    pred.assign(n, -1);
    queue.clear();
    queue.reserve(n);
    queue.push_back(0);
    pred[0] = 0;
    for (size_t head = 0; head < queue.size(); ++head) {
        const int u = queue[head];
        for (int k = start[u]; k < start[u + 1]; ++k)
            if (pred[adj[k]] < 0) {
                pred[adj[k]] = u;
                queue.push_back(adj[k]);
            }
    }
    if ((int)queue.size() != n) return ICPFLOW_E_ARG;   // the edges do not span the points
    std::vector<Edge> &edges = S.edges;
    edges.resize(m);
    for (int e = 0; e < m; ++e) {
        const bool aIsChild = pred[h_edge_a[e]] == h_edge_b[e];
        edges[e] = Edge{h_edge_w[e], aIsChild ? h_edge_b[e] : h_edge_a[e], aIsChild ? h_edge_a[e] : h_edge_b[e]};
    }
    const auto byEnds = [](const Edge &x, const Edge &y) { return x.cur != y.cur ? x.cur < y.cur : x.nxt < y.nxt; };
    bool plain = true;   // non-negative weights order like their bit patterns: radix sort, then order the ties
    bool ascending = true;   // the caller may have sorted by weight already (the GPU does it in passing)
    for (int e = 0; e < m; ++e) {
        if (std::isnan(edges[e].w)) return ICPFLOW_E_ARG;
        plain = plain && !std::signbit(edges[e].w);
        ascending = ascending && (e == 0 || edges[e - 1].w <= edges[e].w);
    }
    const auto order_ties = [&]() {
        for (int i = 0; i < m;) {
            int k = i + 1;
            while (k < m && edges[k].w == edges[i].w) ++k;
            if (k - i > 1) std::sort(edges.begin() + i, edges.begin() + k, byEnds);
            i = k;
        }
    };
    if (ascending) {
        order_ties();
    } else if (plain) {
        std::vector<Edge> &tmp = S.tmp;
        tmp.resize(m);
        std::vector<uint32_t> hist(1 << 16);
        for (int pass = 0; pass < 4; ++pass) {
            const int shift = 16 * pass;
            auto digit = [&](const Edge &x) {
                uint64_t bits;
                std::memcpy(&bits, &x.w, 8);
                return (uint32_t)((bits >> shift) & 0xffff);
            };
            std::fill(hist.begin(), hist.end(), 0u);
            for (const Edge &x : edges) ++hist[digit(x)];
            uint32_t run = 0;
            for (uint32_t &hv : hist) {
                const uint32_t c = hv;
                hv = run;
                run += c;
            }
            for (const Edge &x : edges) tmp[hist[digit(x)]++] = x;
            edges.swap(tmp);
        }
        order_ties();
    } else {
        std::sort(edges.begin(), edges.end(),
                  [&](const Edge &x, const Edge &y) { return x.w != y.w ? x.w < y.w : byEnds(x, y); });
    }

    // 2. single linkage: node n + i = i-th merge
    // (union-find over the POINTS, by size, with path halving; top[root] = the dendrogram node of that component:
    // the same merges as a union-find over the dendrogram nodes, along much shorter paths)
    const int nodes = 2 * n - 1, root = 2 * n - 2;
    std::vector<int> &par = S.par, &top = S.top, &left = S.left, &right = S.right, &count = S.count;
    std::vector<double> &dist = S.dist;
    par.resize(n); top.resize(n); left.resize(m); right.resize(m); dist.resize(m);
    count.assign(nodes, 1);
    std::iota(par.begin(), par.end(), 0);
    std::iota(top.begin(), top.end(), 0);
    auto find = [&](int x) {
        while (par[x] != x) {
            par[x] = par[par[x]];
            x = par[x];
        }
        return x;
    };
    for (int i = 0; i < m; ++i) {
        int ra = find(edges[i].cur), rb = find(edges[i].nxt);
        const int l = top[ra], r = top[rb];
        left[i] = l;
        right[i] = r;
        dist[i] = edges[i].w;
        count[n + i] = count[l] + count[r];
        if (count[l] < count[r]) std::swap(ra, rb);
        par[rb] = ra;
        top[ra] = n + i;
    }

    // 3. condensed tree
    std::vector<Row> &rows = S.rows;
    rows.clear();
    rows.reserve((size_t)n + 64);
    // the points below a dendrogram node, as a contiguous range of one depth-first leaf order (parents carry larger
    // ids than their children: one pass from the root down hands every node its range)
    std::vector<int> &first = S.first, &leafOrder = S.leafOrder;
    first.resize(nodes); leafOrder.resize(n);
    first[root] = 0;
    for (int v = root; v >= n; --v) {
        first[left[v - n]] = first[v];
        first[right[v - n]] = first[v] + count[left[v - n]];
    }
    for (int v = 0; v < n; ++v) leafOrder[first[v]] = v;
    std::vector<int> &relabel = S.relabel, &bfs = S.bfs, &sub = S.sub;
    relabel.assign(nodes, -1);
    bfs.clear();
    sub.clear();
    // breadth first over the dendrogram, but only through nodes that are still part of a cluster: a side that
    // falls out is walked once, by fall_out (the order of the surviving nodes is that of the full walk)
    bfs.reserve(nodes);
    bfs.push_back(root);
    relabel[root] = n;
    int nextLabel = n + 1;
    // every point below `node` leaves the parent (all these rows carry the same lambda, so their order among
    // themselves does not touch the stability sums)
    auto fall_out = [&](int node, int parentLabel, double lambda) {
        const int f = first[node], c = count[node];
        for (int k = f; k < f + c; ++k) rows.push_back(Row{parentLabel, leafOrder[k], lambda, 1});
    };
    for (size_t head = 0; head < bfs.size(); ++head) {
        const int v = bfs[head];
        if (v < n) continue;
        const int l = left[v - n], r = right[v - n];
        const double d = dist[v - n];
        const double lambda = d > 0.0 ? 1.0 / d : std::numeric_limits<double>::infinity();
        const int lc = count[l], rc = count[r];
        if (lc >= min_cluster_size && rc >= min_cluster_size) {
            relabel[l] = nextLabel++;
            rows.push_back(Row{relabel[v], relabel[l], lambda, lc});
            relabel[r] = nextLabel++;
            rows.push_back(Row{relabel[v], relabel[r], lambda, rc});
            bfs.push_back(l);
            bfs.push_back(r);
        } else if (lc < min_cluster_size && rc < min_cluster_size) {
            fall_out(l, relabel[v], lambda);
            fall_out(r, relabel[v], lambda);
        } else if (lc < min_cluster_size) {
            relabel[r] = relabel[v];
            fall_out(l, relabel[v], lambda);
            bfs.push_back(r);
        } else {
            relabel[l] = relabel[v];
            fall_out(r, relabel[v], lambda);
            bfs.push_back(l);
        }
    }

    // 4. stability per cluster id (ids n .. nextLabel - 1; n is the root)
    const int clusters = nextLabel - n;
    std::vector<double> birth(nextLabel, std::numeric_limits<double>::quiet_NaN()), stability(clusters, 0.0);
    for (const Row &w : rows) birth[w.child] = w.lambda;
    birth[n] = 0.0;
    for (const Row &w : rows) stability[w.parent - n] += (w.lambda - birth[w.parent]) * (double)w.size;

    // 5. excess of mass.  Cluster children of a cluster come in pairs and carry larger ids than their parent.
    std::vector<int> childA(clusters, -1), childB(clusters, -1);
    for (const Row &w : rows)
        if (w.size > 1) (childA[w.parent - n] < 0 ? childA[w.parent - n] : childB[w.parent - n]) = w.child - n;
    std::vector<uint8_t> selected(clusters, 1);
    selected[0] = 0;   // the root is no candidate (allow_single_cluster False)
    for (int c = clusters - 1; c >= 1; --c) {
        double below = 0.0;
        if (childA[c] >= 0) below = stability[childA[c]] + (childB[c] >= 0 ? stability[childB[c]] : 0.0);
        if (below > stability[c]) {
            selected[c] = 0;
            stability[c] = below;
        } else {   // the cluster stands: nothing beneath it does
            sub.clear();
            if (childA[c] >= 0) sub.push_back(childA[c]);
            if (childB[c] >= 0) sub.push_back(childB[c]);
            for (size_t head = 0; head < sub.size(); ++head) {
                const int x = sub[head];
                selected[x] = 0;
                if (childA[x] >= 0) sub.push_back(childA[x]);
                if (childB[x] >= 0) sub.push_back(childB[x]);
            }
        }
    }

    // 6. labels: rank of the selected ids; a point follows its nearest selected ancestor
    std::vector<int> label(clusters, -1), owner(clusters, -1);
    int next = 0;
    for (int c = 1; c < clusters; ++c)
        if (selected[c]) label[c] = next++;
    for (const Row &w : rows) {   // rows are top-down: a cluster's parent row precedes its own rows
        const int p = w.parent - n;
        if (w.size > 1) {
            const int c = w.child - n;
            owner[c] = selected[c] ? c : owner[p];
        } else {
            const int o = (p == 0) ? -1 : (selected[p] ? p : owner[p]);
            h_labels[w.child] = o < 0 ? -1 : label[o];
        }
    }
    return 0;
}

/*
 * poa_oracle.c -- CPU statement of the partial-order aligner behind spoa.poa(...) and inside find_consensus.
 *
 * TEST INFRASTRUCTURE ONLY (same rules as ssw_oracle.c): nothing under ciri_long_amd/ may link, load or call this.
 *
 * PARITY UNPINNED.  CIRI-long calls the external package pyspoa (`from spoa import poa`, requirement >= 0.0.5,
 * setup.py:57): collapse.py:267,504 `poa(seqs, 2, False, 10, -4, -8, -2, -24, -1)`, tests/test_poa.py:30
 * `poa(seqs, 0, True, 10, -4, -8, -2, -24, -1)`.  Neither pyspoa nor the spoa C++ library it wraps exists in
 * /root/reference or in this environment, so there is no file to follow line by line and no output to compare with.
 * This file restates the PUBLISHED algorithm -- spoa (Vaser, Sovic, Nagarajan, Sikic: Genome Res. 27:737-746, 2017;
 * github.com/rvaser/spoa, the 4.0 line pyspoa bundles), itself an implementation of Lee, Grasso, Sharlow:
 * Bioinformatics 18:452-464 (2002) -- from the author's knowledge of that code base:
 * `AlignmentEngine::Create` (sub-type selection), `SisdAlignmentEngine::Initialize/Linear/Affine/Convex`
 * (recurrences, end cell, back-track order), `Graph::AddAlignment` (node numbering, aligned lists, edge weights),
 * `Graph::TopologicalSort` (the depth-first order every alignment and the consensus run in),
 * `Graph::TraverseHeaviestBundle/BranchCompletion`, `Graph::GenerateMultipleSequenceAlignment`.  Every rule below that
 * could not be checked against the sources here is a recollection: [UNVERIFIED] applies to the whole file.
 * "clh-poa v3": the three departures of v2 (five letter codes, incremental rank rule, node ids in sequence order) are
 * gone -- letters are raw bytes, node ids follow AddAlignment, the rank order is spoa's depth-first sort after every
 * sequence.
 *
 * Model
 * -----
 * poa(seqs, algorithm, genmsa, m, n, g, e, q, c): sequences are added one by one: align to the graph, fuse the path,
 * sort the graph.  An empty sequence is skipped (it gets no MSA row either).
 * algorithm: 0 local (Smith-Waterman), 1 global (Needleman-Wunsch), 2 overlap (sequence end to end, graph ends free).
 * m match, n mismatch (n < 0), gap of k bases costs max(g + (k-1) e, q + (k-1) c) (all <= 0).
 * Sub-type as spoa selects it: g >= e -> linear (e := g); else g <= q or e >= c -> affine (one piece); else convex.
 * Letters: bytes.  Two letters match iff the bytes are equal (spoa numbers the characters in order of first appearance and
 * compares the numbers): 'a' is not 'A', an N matches an N, every IUPAC letter is a letter of its own.
 *
 * Alignment of a sequence s[1..L] to the graph, rows i = 1..N in topological order ("rank"), row 0 = no node:
 *   pred(i) = in-edges of the row's node in insertion order; a node without in-edges has the single predecessor row 0.
 *   F[i][j] = max over pred p of max(H[p][j] + g, F[p][j] + e)        O[i][j] = same with q, c       (gap in the sequence)
 *   E[i][j] = max(H[i][j-1] + g, E[i][j-1] + e)                       Q[i][j] = same with q, c       (gap in the graph)
 *   H[i][j] = max(max over p of H[p][j-1] + s(i, j), F, E, O, Q)      local: also >= 0
 *   borders: E[0][j] = g + (j-1) e, Q[0][j] = q + (j-1) c, F[0][j] = O[0][j] = -inf (j >= 1);
 *            F[i][0] = e + max over p of F[p][0] (a source row: g), O likewise with q, c; E[i][0] = Q[i][0] = -inf;
 *            H[0][0] = 0; local: H[0][j] = H[i][0] = 0; global: H[0][j] = max(E, Q)[0][j], H[i][0] = max(F, O)[i][0];
 *            overlap: H[0][j] = max(E, Q)[0][j], H[i][0] = 0.
 *   end cell = first strict maximum in (rank, column) order over: local all cells (initial maximum 0: nothing positive
 *   -> empty alignment); global cells (sink row, L); overlap cells (sink row, any column).
 *   Back-track from the end cell until: local H == 0; global (0, 0); overlap i == 0 or j == 0.  At (i, j), first hit of
 *     1. diagonal: H == H[p][j-1] + s(i, j), p in in-edge order;
 *     2. vertical, p in in-edge order, per p in this order: H == F[p][j] + e (extend-up), H == H[p][j] + g,
 *        H == O[p][j] + c (extend-up), H == H[p][j] + q;
 *     3. horizontal: H == E[i][j-1] + e (extend-left), H == H[i][j-1] + g, H == Q[i][j-1] + c (extend-left),
 *        H == H[i][j-1] + q;
 *   one step is emitted and taken; then with extend-left, further horizontal steps are taken, stopping after the step
 *   that arrives at a column j with neither E[i][j] + e == E[i][j+1] nor Q[i][j] + c == Q[i][j+1]; with extend-up,
 *   further vertical steps: at (i, j) the first p with F[i][j] == H[p][j] + g (last step), F[i][j] == F[p][j] + e,
 *   O[i][j] == H[p][j] + q (last step), O[i][j] == O[p][j] + c is taken, ending after a last step or in row 0.
 *   Affine: the same without O and Q.  Linear: H only; diagonal, then vertical (H == H[p][j] + g), then horizontal.
 *   The alignment is the list of pairs (node | none, base | none) of the steps.  An alignment that holds steps but no base
 *   (overlap mode: a path of vertical steps only) makes AddAlignment throw; here: return -3.
 *
 * Fusing (Graph::AddAlignment).  The bases of the alignment are a contiguous stretch [jb, je] of the sequence.  Node ids
 * are given in this order: the bases in front of jb (a chain of new nodes), the bases behind je (another chain), then the
 * stretch itself base by base -- a base aligned to a node with the same letter re-uses it; with another letter it re-uses
 * the first member of that node's aligned list holding its letter, else a new node joins the set (every member's list
 * gains the new node; the new node's list is the matched node's list followed by the matched node); a base without a
 * node (insertion) gets a new node.  An empty alignment: the whole sequence is one chain of new nodes.  Consecutive
 * bases get an edge (an existing edge gains weight and the sequence's label; a new edge goes to the end of the in-edge
 * list of its head); with unit base weights one sequence contributes 2 per edge.
 *
 * Rank order (Graph::TopologicalSort, after every sequence).  Iterative depth-first search over the node ids in ascending
 * order with an explicit stack: a node on top of the stack pushes its in-edge tails that are not finished (in list order),
 * then -- unless it was itself pushed as an aligned node ("ignored") -- the unfinished members of its aligned list (in list
 * order, each flagged ignored); if it pushed nothing it is finished and, unless ignored, emitted followed by its whole
 * aligned list in list order.  So an aligned set is contiguous in the order, led by the member the search met first.
 *
 * Consensus (Graph::TraverseHeaviestBundle): in rank order, a node takes the in-edge with the largest weight, on
 * equal weight the later edge if the score of its tail is not smaller; score = weight + score(tail), a node without
 * in-edges scores -1 (spoa's initial value); the best node is the first with the strictly largest score.  While the best
 * node has out-edges (BranchCompletion): the other tails of its successors are barred (score -1), the scores of all later
 * ranks are recomputed (barred tails skipped; a node all of whose tails are barred becomes barred) and the best later
 * node is taken.  The path is followed back over the chosen edges.
 * min_coverage > 0 (GenerateConsensus(min_coverage) of later spoa releases; pyspoa 0.0.5's poa() has no such argument
 * and passes 0 here; find_consensus uses it, see ccs_oracle.c): nodes of the path whose coverage -- the number of
 * sequences with an edge at the node (Node::Coverage: a one-base sequence has no edge) -- is smaller are left out.
 * MSA (GenerateMultipleSequenceAlignment): one column per aligned set in rank order, one row per sequence, '-' elsewhere.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define NEG (-(1 << 29))

typedef struct { int32_t *v; int n, cap; } ivec;
static void iv_push(ivec *a, int32_t x)
{
    if (a->n == a->cap) { a->cap = a->cap ? 2 * a->cap : 4; a->v = (int32_t *)realloc(a->v, sizeof(int32_t) * (size_t)a->cap); }
    a->v[a->n++] = x;
}

typedef struct {
    int n, cap;
    uint8_t *code;
    ivec *tail, *wt;          /* in-edges in insertion order: tail node, weight */
    ivec *al;                 /* aligned_nodes, in list order */
    int32_t *nout;
    int32_t *cov;             /* Node::Coverage: sequences with an edge at the node */
    int32_t *order, *rank;    /* order[r-1] = node of rank r; rank[node] = 1..n */
} graph;

typedef struct { int algorithm, m, n, g, e, q, c, subtype, min_cov; } par;   /* subtype 0 linear, 1 affine, 2 convex */

static inline int imax(int a, int b) { return a > b ? a : b; }

static void g_init(graph *g, int cap)
{
    g->n = 0; g->cap = cap;
    g->code = (uint8_t *)malloc((size_t)cap);
    g->tail = (ivec *)calloc((size_t)cap, sizeof(ivec)); g->wt = (ivec *)calloc((size_t)cap, sizeof(ivec)); g->al = (ivec *)calloc((size_t)cap, sizeof(ivec));
    g->nout = (int32_t *)calloc((size_t)cap, sizeof(int32_t));
    g->cov = (int32_t *)calloc((size_t)cap, sizeof(int32_t));
    g->order = (int32_t *)malloc(sizeof(int32_t) * (size_t)cap);
    g->rank = (int32_t *)malloc(sizeof(int32_t) * (size_t)cap);
}
static void g_free(graph *g)
{
    for (int v = 0; v < g->cap; ++v) { free(g->tail[v].v); free(g->wt[v].v); free(g->al[v].v); }
    free(g->code); free(g->tail); free(g->wt); free(g->al); free(g->nout); free(g->cov); free(g->order); free(g->rank);
}
static int g_new(graph *g, int code)                      /* Graph::AddNode */
{
    if (g->n >= g->cap) return -1;
    const int v = g->n++;
    g->code[v] = (uint8_t)code; g->rank[v] = 0;
    return v;
}
static void g_edge(graph *g, int u, int v, int w)         /* Graph::AddEdge */
{
    for (int k = 0; k < g->tail[v].n; ++k)
        if (g->tail[v].v[k] == u) { g->wt[v].v[k] += w; return; }
    iv_push(&g->tail[v], u); iv_push(&g->wt[v], w);
    g->nout[u] += 1;
}
static int max_indegree(const graph *g)
{
    int m = 1;
    for (int v = 0; v < g->n; ++v) m = imax(m, g->tail[v].n);
    return m;
}
/* rows of the predecessors of row i (1..N): count, list; a source row has the single predecessor 0 */
static int preds_of(const graph *g, int i, int *pr)
{
    const int v = g->order[i - 1];
    if (g->tail[v].n == 0) { pr[0] = 0; return 1; }
    for (int k = 0; k < g->tail[v].n; ++k) pr[k] = g->rank[g->tail[v].v[k]];
    return g->tail[v].n;
}

/* Graph::TopologicalSort */
static void topo_sort(graph *g)
{
    const int N = g->n;
    uint8_t *marks = (uint8_t *)calloc((size_t)N + 1, 1), *ignored = (uint8_t *)calloc((size_t)N + 1, 1);   /* 0 unmarked, 1 temporarily, 2 permanently marked */
    ivec st = {0, 0, 0};
    int nord = 0;
    for (int i = 0; i < N; ++i) {
        if (marks[i] != 0) continue;
        iv_push(&st, i);
        while (st.n) {
            const int curr = st.v[st.n - 1];
            int valid = 1;
            if (marks[curr] != 2) {
                for (int k = 0; k < g->tail[curr].n; ++k) {
                    const int t = g->tail[curr].v[k];
                    if (marks[t] != 2) { iv_push(&st, t); valid = 0; }
                }
                if (!ignored[curr])
                    for (int k = 0; k < g->al[curr].n; ++k) {
                        const int a = g->al[curr].v[k];
                        if (marks[a] != 2) { iv_push(&st, a); ignored[a] = 1; valid = 0; }
                    }
                if (valid) {
                    marks[curr] = 2;
                    if (!ignored[curr]) {
                        g->order[nord++] = curr;
                        for (int k = 0; k < g->al[curr].n; ++k) g->order[nord++] = g->al[curr].v[k];
                    }
                } else marks[curr] = 1;
            }
            if (valid) --st.n;
        }
    }
    for (int r = 0; r < nord; ++r) g->rank[g->order[r]] = r + 1;
    free(marks); free(ignored); free(st.v);
}

/* ------------------------------------------------------------------------------------------------------------------
 * alignment.  pn[j] (j = 0..L-1) = node the base j is aligned to, or -1.  [*jb, *je] = the bases the alignment holds
 * (jb > je: none), *steps = its number of pairs.  Pairs (node, no base) do not change the graph (AddAlignment skips them)
 * and are only counted.  *score = end-cell score. */
static void align_linear(const graph *g, const uint8_t *s, int L, const par *P, int32_t *pn, int *score, int *jb, int *je, int *steps)
{
    const int N = g->n, W = L + 1, sw = P->algorithm == 0, nw = P->algorithm == 1;
    int32_t *H = (int32_t *)malloc(sizeof(int32_t) * (size_t)(N + 1) * W);
    int *pr = (int *)malloc(sizeof(int) * (size_t)(max_indegree(g) + 1));
    H[0] = 0;
    for (int j = 1; j <= L; ++j) H[j] = sw ? 0 : j * P->g;
    for (int i = 1; i <= N; ++i) {
        if (!nw) { H[(size_t)i * W] = 0; continue; }
        const int np = preds_of(g, i, pr);
        int pen = NEG;
        for (int k = 0; k < np; ++k) pen = imax(pen, H[(size_t)pr[k] * W]);
        H[(size_t)i * W] = pen + P->g;
    }
    int best = sw ? 0 : NEG, bi = 0, bj = 0;
    for (int i = 1; i <= N; ++i) {
        const int v = g->order[i - 1], np = preds_of(g, i, pr), sink = g->nout[v] == 0;
        int32_t *Hr = H + (size_t)i * W;
        for (int j = 1; j <= L; ++j) {
            const int sc = g->code[v] == s[j - 1] ? P->m : P->n;
            int h = NEG;
            for (int k = 0; k < np; ++k) h = imax(h, imax(H[(size_t)pr[k] * W + j - 1] + sc, H[(size_t)pr[k] * W + j] + P->g));
            h = imax(h, Hr[j - 1] + P->g);
            if (sw) h = imax(h, 0);
            Hr[j] = h;
            if ((sw || (nw && sink && j == L) || (P->algorithm == 2 && sink)) && best < h) { best = h; bi = i; bj = j; }
        }
    }
    for (int j = 0; j < L; ++j) pn[j] = -1;
    *score = best;
    int i = bi, j = bj, ns = 0;
    while ((sw && H[(size_t)i * W + j] != 0) || (nw && (i != 0 || j != 0)) || (P->algorithm == 2 && i != 0 && j != 0)) {
        const int h = H[(size_t)i * W + j];
        int pi = i, pj = j, found = 0;
        if (i != 0 && j != 0) {
            const int v = g->order[i - 1], np = preds_of(g, i, pr), sc = g->code[v] == s[j - 1] ? P->m : P->n;
            for (int k = 0; k < np && !found; ++k) if (h == H[(size_t)pr[k] * W + j - 1] + sc) { pi = pr[k]; pj = j - 1; found = 1; }
        }
        if (!found && i != 0) {
            const int np = preds_of(g, i, pr);
            for (int k = 0; k < np && !found; ++k) if (h == H[(size_t)pr[k] * W + j] + P->g) { pi = pr[k]; pj = j; found = 1; }
        }
        if (!found && j != 0 && h == H[(size_t)i * W + j - 1] + P->g) { pi = i; pj = j - 1; found = 1; }
        if (!found) break;                                   /* cannot happen */
        if (pj != j && pi != i) pn[j - 1] = g->order[i - 1];
        ++ns;
        i = pi; j = pj;
    }
    *jb = j; *je = bj - 1; *steps = ns;
    free(H); free(pr);
}

static void align_gotoh(const graph *g, const uint8_t *s, int L, const par *P, int32_t *pn, int *score, int *jb, int *je, int *steps)
{
    const int N = g->n, W = L + 1, sw = P->algorithm == 0, nw = P->algorithm == 1, ov = P->algorithm == 2, cx = P->subtype == 2;
    const int ge = P->g, ee = P->e, qq = P->q, cc = P->c;
    const size_t cells = (size_t)(N + 1) * W;
    int32_t *H = (int32_t *)malloc(sizeof(int32_t) * cells), *F = (int32_t *)malloc(sizeof(int32_t) * cells), *E = (int32_t *)malloc(sizeof(int32_t) * cells);
    int32_t *O = (int32_t *)malloc(sizeof(int32_t) * cells), *Q = (int32_t *)malloc(sizeof(int32_t) * cells);
    int *pr = (int *)malloc(sizeof(int) * (size_t)(max_indegree(g) + 1));
    /* Initialize */
    O[0] = 0; Q[0] = 0; F[0] = 0; E[0] = 0; H[0] = 0;
    for (int j = 1; j <= L; ++j) { O[j] = NEG; Q[j] = cx ? qq + (j - 1) * cc : NEG; F[j] = NEG; E[j] = ge + (j - 1) * ee; }
    for (int i = 1; i <= N; ++i) {
        const int v = g->order[i - 1];
        int penF = g->tail[v].n == 0 ? ge - ee : NEG, penO = g->tail[v].n == 0 ? qq - cc : NEG;
        for (int k = 0; k < g->tail[v].n; ++k) {
            const int p = g->rank[g->tail[v].v[k]];
            penF = imax(penF, F[(size_t)p * W]); penO = imax(penO, O[(size_t)p * W]);
        }
        F[(size_t)i * W] = penF + ee; E[(size_t)i * W] = NEG;
        O[(size_t)i * W] = cx ? penO + cc : NEG; Q[(size_t)i * W] = NEG;
    }
    for (int j = 1; j <= L; ++j) H[j] = sw ? 0 : imax(Q[j], E[j]);
    for (int i = 1; i <= N; ++i) H[(size_t)i * W] = nw ? imax(O[(size_t)i * W], F[(size_t)i * W]) : 0;
    /* rows */
    int best = sw ? 0 : NEG, bi = 0, bj = 0;
    for (int i = 1; i <= N; ++i) {
        const int v = g->order[i - 1], np = preds_of(g, i, pr), sink = g->nout[v] == 0;
        int32_t *Hr = H + (size_t)i * W, *Fr = F + (size_t)i * W, *Er = E + (size_t)i * W, *Or = O + (size_t)i * W, *Qr = Q + (size_t)i * W;
        for (int j = 1; j <= L; ++j) {
            const int sc = g->code[v] == s[j - 1] ? P->m : P->n;
            int f = NEG, o = NEG, h = NEG;
            for (int k = 0; k < np; ++k) {
                const size_t b = (size_t)pr[k] * W;
                f = imax(f, imax(H[b + j] + ge, F[b + j] + ee));
                if (cx) o = imax(o, imax(H[b + j] + qq, O[b + j] + cc));
                h = imax(h, H[b + j - 1] + sc);
            }
            Fr[j] = f; Or[j] = o; Hr[j] = h;
        }
        for (int j = 1; j <= L; ++j) {
            Er[j] = imax(Hr[j - 1] + ge, Er[j - 1] + ee);
            Qr[j] = cx ? imax(Hr[j - 1] + qq, Qr[j - 1] + cc) : NEG;
            int h = imax(Hr[j], imax(imax(Fr[j], Er[j]), imax(Or[j], Qr[j])));
            if (sw) h = imax(h, 0);
            Hr[j] = h;
            if ((sw || (nw && sink && j == L) || (ov && sink)) && best < h) { best = h; bi = i; bj = j; }
        }
    }
    for (int j = 0; j < L; ++j) pn[j] = -1;
    *score = best;
    /* back-track */
    int i = bi, j = bj, ns = 0;
    while ((sw && H[(size_t)i * W + j] != 0) || (nw && (i != 0 || j != 0)) || (ov && i != 0 && j != 0)) {
        const int h = H[(size_t)i * W + j];
        int pi = i, pj = j, found = 0, ext_left = 0, ext_up = 0;
        if (i != 0 && j != 0) {
            const int v = g->order[i - 1], np = preds_of(g, i, pr), sc = g->code[v] == s[j - 1] ? P->m : P->n;
            for (int k = 0; k < np && !found; ++k) if (h == H[(size_t)pr[k] * W + j - 1] + sc) { pi = pr[k]; pj = j - 1; found = 1; }
        }
        if (!found && i != 0) {
            const int np = preds_of(g, i, pr);
            for (int k = 0; k < np && !found; ++k) {
                const size_t b = (size_t)pr[k] * W + j;
                if (h == F[b] + ee) { ext_up = 1; found = 1; }
                else if (h == H[b] + ge) found = 1;
                else if (cx && h == O[b] + cc) { ext_up = 1; found = 1; }
                else if (cx && h == H[b] + qq) found = 1;
                if (found) { pi = pr[k]; pj = j; }
            }
        }
        if (!found && j != 0) {
            const size_t b = (size_t)i * W + j - 1;
            if (h == E[b] + ee) { ext_left = 1; found = 1; }
            else if (h == H[b] + ge) found = 1;
            else if (cx && h == Q[b] + cc) { ext_left = 1; found = 1; }
            else if (cx && h == H[b] + qq) found = 1;
            if (found) { pi = i; pj = j - 1; }
        }
        if (!found) break;                                   /* cannot happen */
        if (pi != i && pj != j) pn[j - 1] = g->order[i - 1];
        ++ns;
        i = pi; j = pj;
        if (ext_left) {
            for (;;) {
                --j; ++ns;                                    /* one more base without a node */
                const size_t b = (size_t)i * W + j;
                if (E[b] + ee != E[b + 1] && (!cx || Q[b] + cc != Q[b + 1])) break;
            }
        } else if (ext_up) {
            for (;;) {
                int stop = 1, ni = 0;
                const int np = preds_of(g, i, pr);
                const size_t a = (size_t)i * W + j;
                for (int k = 0; k < np; ++k) {
                    const size_t b = (size_t)pr[k] * W + j;
                    int hit = 0;
                    if (F[a] == H[b] + ge) { stop = 1; hit = 1; }
                    else if (F[a] == F[b] + ee) { stop = 0; hit = 1; }
                    else if (cx && O[a] == H[b] + qq) { stop = 1; hit = 1; }
                    else if (cx && O[a] == O[b] + cc) { stop = 0; hit = 1; }
                    else stop = 0;
                    if (hit) { ni = pr[k]; break; }
                }
                i = ni; ++ns;                                 /* one more node without a base */
                if (stop || i == 0) break;
            }
        }
    }
    *jb = j; *je = bj - 1; *steps = ns;
    free(H); free(F); free(E); free(O); free(Q); free(pr);
}

/* ------------------------------------------------------------------------------------------------------------------
 * Graph::AddAlignment.  used[j] receives the node of base j.  The alignment holds the bases [jb, je] (steps == 0: it is
 * empty).  returns 0, -1 (node capacity), -3 (an alignment without a base: spoa throws). */
static int fuse(graph *g, const uint8_t *s, int L, const int32_t *pn, int jb, int je, int steps, int32_t *used)
{
    if (steps == 0) { jb = L; je = L - 1; }                   /* AddSequence(0, L) */
    else if (jb > je) return -3;
    /* the bases in front of the alignment, then the bases behind it: two chains of new nodes (Graph::AddSequence) */
    for (int j = 0; j < jb; ++j) {
        if ((used[j] = g_new(g, s[j])) < 0) return -1;
        if (j > 0) g_edge(g, used[j - 1], used[j], 2);
    }
    if (steps == 0) return 0;
    for (int j = je + 1; j < L; ++j) {
        if ((used[j] = g_new(g, s[j])) < 0) return -1;
        if (j > je + 1) g_edge(g, used[j - 1], used[j], 2);
    }
    for (int j = jb; j <= je; ++j) {
        const int b = s[j], v = pn[j];
        int use = -1;
        if (v < 0) use = g_new(g, b);
        else if (g->code[v] == b) use = v;
        else {
            for (int k = 0; k < g->al[v].n; ++k) { const int w = g->al[v].v[k]; if (g->code[w] == b) { use = w; break; } }
            if (use < 0) {
                use = g_new(g, b);
                if (use < 0) return -1;
                for (int k = 0; k < g->al[v].n; ++k) {
                    const int w = g->al[v].v[k];
                    iv_push(&g->al[w], use); iv_push(&g->al[use], w);
                }
                iv_push(&g->al[v], use); iv_push(&g->al[use], v);
            }
        }
        if (use < 0) return -1;
        used[j] = use;
        if (j > 0) g_edge(g, used[j - 1], use, 2);          /* j == jb: from the last node of the leading chain, if there is one */
    }
    if (je + 1 < L) g_edge(g, used[je], used[je + 1], 2);
    return 0;
}

/* ------------------------------------------------------------------------------------------------------------------ */
static int consensus(const graph *g, int32_t *path, int cap)
{
    const int N = g->n;
    if (N == 0) return 0;
    int64_t *score = (int64_t *)malloc(sizeof(int64_t) * (size_t)N);
    int32_t *bp = (int32_t *)malloc(sizeof(int32_t) * (size_t)N);
    for (int v = 0; v < N; ++v) { score[v] = -1; bp[v] = -1; }
    int top = -1;
    for (int r = 1; r <= N; ++r) {
        const int v = g->order[r - 1];
        for (int k = 0; k < g->tail[v].n; ++k) {
            const int u = g->tail[v].v[k], w = g->wt[v].v[k];
            if (score[v] < w || (score[v] == w && score[bp[v]] <= score[u])) { score[v] = w; bp[v] = u; }
        }
        if (bp[v] >= 0) score[v] += score[bp[v]];
        if (top < 0 || score[top] < score[v]) top = v;
    }
    while (g->nout[top] != 0) {                                /* BranchCompletion */
        const int start = top, r0 = g->rank[start];
        for (int h = 0; h < N; ++h) {
            int is_succ = 0;
            for (int k = 0; k < g->tail[h].n; ++k) if (g->tail[h].v[k] == start) is_succ = 1;
            if (!is_succ) continue;
            for (int k = 0; k < g->tail[h].n; ++k) if (g->tail[h].v[k] != start) score[g->tail[h].v[k]] = -1;
        }
        top = -1;
        for (int r = r0 + 1; r <= N; ++r) {
            const int v = g->order[r - 1];
            score[v] = -1; bp[v] = -1;
            for (int k = 0; k < g->tail[v].n; ++k) {
                const int u = g->tail[v].v[k], w = g->wt[v].v[k];
                if (score[u] == -1) continue;
                if (score[v] < w || (score[v] == w && score[bp[v]] <= score[u])) { score[v] = w; bp[v] = u; }
            }
            if (bp[v] >= 0) score[v] += score[bp[v]];
            if (top < 0 || score[top] < score[v]) top = v;
        }
        if (top < 0) { top = start; break; }                   /* cannot happen: a node with out-edges is not last */
    }
    int len = 0;
    for (int v = top; v >= 0; v = bp[v]) ++len;
    if (len > cap) { free(score); free(bp); return -1; }
    int k = len;
    for (int v = top; v >= 0; v = bp[v]) path[--k] = v;
    free(score); free(bp);
    return len;
}

static int parse_par(const int32_t *pp, par *P)
{
    P->algorithm = pp[0]; P->m = pp[1]; P->n = pp[2]; P->g = pp[3]; P->e = pp[4]; P->q = pp[5]; P->c = pp[6]; P->min_cov = pp[7];
    if (P->algorithm < 0 || P->algorithm > 2) return -2;
    if (P->g > 0 || P->q > 0 || P->e > 0 || P->c > 0) return -2;
    P->subtype = P->g >= P->e ? 0 : ((P->g <= P->q || P->e >= P->c) ? 1 : 2);      /* AlignmentEngine::Create */
    if (P->subtype == 0) P->e = P->g;
    else if (P->subtype == 1) { P->q = P->g; P->c = P->e; }
    return 0;
}

/* poa(seqs, algorithm, genmsa, m, n, g, e, q, c): nseq sequences (packed letters = bytes, off[nseq+1]); params = {algorithm,
 * m, n, g, e, q, c, min_coverage}.  cons receives the consensus letters (capacity cap); returns its length, -1 on an
 * implementation limit, -2 on invalid parameters, -3 where spoa throws (an alignment without a base).  msa (may be NULL)
 * receives one row of *ncols characters ('-' = 45, else the letter) per NON-EMPTY sequence, row-major, if it fits msa_cap
 * (else -1).  scores (may be NULL) receives the end-cell score of every alignment (0 for an empty sequence).
 * rank_out (may be NULL): node ids in the final rank order, preceded by their count (capacity: total letters + 1). */
int clo_poa_ranked(int32_t nseq, const int8_t *seqs_, const int32_t *off, const int32_t *params, int8_t *cons, int32_t cap,
                   int8_t *msa, int64_t msa_cap, int32_t *ncols, int32_t *scores, int32_t *rank_out)
{
    const uint8_t *seqs = (const uint8_t *)seqs_;
    par P;
    if (parse_par(params, &P) != 0) return -2;
    const int total = off[nseq];
    graph g;
    g_init(&g, total + 8);
    int32_t *pn = (int32_t *)malloc(sizeof(int32_t) * (size_t)(total + 1)), *used = (int32_t *)malloc(sizeof(int32_t) * (size_t)(total + 1));
    int rc = 0;
    for (int s = 0; s < nseq && rc == 0; ++s) {
        const int L = off[s + 1] - off[s];
        if (scores) scores[s] = 0;
        if (L == 0) continue;
        int sc = 0, jb = 0, je = -1, steps = 0;
        if (g.n == 0) { for (int j = 0; j < L; ++j) pn[j] = -1; }
        else if (P.subtype == 0) align_linear(&g, seqs + off[s], L, &P, pn, &sc, &jb, &je, &steps);
        else align_gotoh(&g, seqs + off[s], L, &P, pn, &sc, &jb, &je, &steps);
        if (scores) scores[s] = sc;
        rc = fuse(&g, seqs + off[s], L, pn, jb, je, steps, used + off[s]);
        if (rc == 0) {
            if (L >= 2) for (int j = 0; j < L; ++j) g.cov[used[off[s] + j]] += 1;
            topo_sort(&g);
        }
    }
    int len = rc;
    if (rc == 0) {
        int32_t *path = (int32_t *)malloc(sizeof(int32_t) * (size_t)(g.n + 1));
        len = consensus(&g, path, g.n + 1);
        if (len > cap) len = -1;
        if (len >= 0) {                                        /* GenerateConsensus(min_coverage): nodes below it are left out */
            int k2 = 0;
            for (int k = 0; k < len; ++k) if (g.cov[path[k]] >= P.min_cov) cons[k2++] = (int8_t)g.code[path[k]];
            len = k2;
        }
        free(path);
    }
    if (len >= 0 && rank_out) { rank_out[0] = g.n; for (int i = 0; i < g.n; ++i) rank_out[1 + i] = g.order[i]; }
    if (len >= 0 && ncols) {
        int32_t *col = (int32_t *)malloc(sizeof(int32_t) * (size_t)(g.n + 1));
        int nc = 0;
        for (int i = 0; i < g.n; ++i, ++nc) {
            const int v = g.order[i];
            col[v] = nc;
            for (int k = 0; k < g.al[v].n; ++k) { col[g.al[v].v[k]] = nc; ++i; }
        }
        *ncols = nc;
        if (msa) {
            int rows = 0;
            for (int s = 0; s < nseq; ++s) rows += off[s + 1] > off[s];
            if ((int64_t)rows * nc > msa_cap) len = -1;
            else {
                memset(msa, '-', (size_t)rows * nc);
                int r = 0;
                for (int s = 0; s < nseq; ++s) {
                    if (off[s + 1] == off[s]) continue;
                    for (int j = off[s]; j < off[s + 1]; ++j) msa[(size_t)r * nc + col[used[j]]] = (int8_t)g.code[used[j]];
                    ++r;
                }
            }
        }
        free(col);
    }
    free(pn); free(used);
    g_free(&g);
    return len;
}

int clo_poa(int32_t nseq, const int8_t *seqs, const int32_t *off, const int32_t *params, int8_t *cons, int32_t cap,
            int8_t *msa, int64_t msa_cap, int32_t *ncols, int32_t *scores)
{
    return clo_poa_ranked(nseq, seqs, off, params, cons, cap, msa, msa_cap, ncols, scores, NULL);
}

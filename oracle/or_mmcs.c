/* TEST INFRASTRUCTURE -- CPU oracle (see oracle.h header: parity unpinned).
 *
 * Plonky3's MerkleTreeMmcs (p3-merkle-tree MerkleTree::new / first_digest_layer / compress_and_inject and
 * MerkleTreeMmcs::verify_batch; RECALLED -- SP1 reaches the crate from reference
 * provers/sp1/driver/src/lib.rs:48-57, the source is outside the tree) on the oracle's configured
 * Poseidon2 sponge (or_hash_elem_slice) and 2-to-1 compression (or_hash_pair).  Matrices of power-of-two
 * heights, each row-major or column-major; digests in heap order (leaves at H + i, root at 1). */
#include "oracle.h"
#include <stdlib.h>
#include <string.h>

static fp mat_at(const or_matrix* m, size_t r, size_t c) {
    return m->row_major ? m->values[r * m->width + c] : m->values[c * m->height + r];
}
/* hash of the concatenated rows `r` of every matrix of height h; 0 when there is none */
static int level_hash(const or_matrix* mats, uint32_t n, uint32_t h, size_t r, uint32_t* digest) {
    size_t tot = 0;
    for (uint32_t m = 0; m < n; m++) if (mats[m].height == h) tot += mats[m].width;
    if (!tot) return 0;
    fp* cat = (fp*)malloc(tot * sizeof(fp));
    size_t pos = 0;
    for (uint32_t m = 0; m < n; m++)
        if (mats[m].height == h)
            for (size_t c = 0; c < mats[m].width; c++) cat[pos++] = mat_at(&mats[m], r, c);
    or_hash_elem_slice(cat, tot, 1, digest);
    free(cat);
    return 1;
}
static uint32_t max_height(const or_matrix* mats, uint32_t n) {
    uint32_t H = 0;
    for (uint32_t m = 0; m < n; m++) if (mats[m].height > H) H = mats[m].height;
    return H;
}

void or_mmcs_commit(const or_matrix* mats, uint32_t n, uint32_t* nodes) {
    uint32_t H = max_height(mats, n);
    for (size_t i = 0; i < H; i++) level_hash(mats, n, H, i, nodes + (H + i) * 8);
    for (uint32_t size = H / 2; size >= 1; size /= 2)
        for (size_t i = 0; i < size; i++) {
            uint32_t* node = nodes + (size + i) * 8;
            uint32_t pair[8], extra[8];
            or_hash_pair(nodes + 2 * (size + i) * 8, nodes + (2 * (size + i) + 1) * 8, pair);
            if (level_hash(mats, n, size, i, extra)) or_hash_pair(pair, extra, node);
            else memcpy(node, pair, 32);
        }
}

/* rows: concatenated opened rows in commit order; path: log2(H) siblings from the leaf level up */
int or_mmcs_verify(const uint32_t* heights, const uint32_t* widths, uint32_t n, uint32_t index, const fp* rows, const uint32_t* path,
                   const uint32_t* root) {
    or_matrix* one = (or_matrix*)calloc(n ? n : 1, sizeof(or_matrix)); /* every matrix as its single opened row */
    uint32_t H = 0;
    size_t pos = 0;
    for (uint32_t m = 0; m < n; m++) {
        one[m].values = rows + pos; one[m].height = heights[m]; one[m].width = widths[m]; one[m].row_major = 1;
        pos += widths[m];
        if (heights[m] > H) H = heights[m];
    }
    /* level_hash indexes row r of a row-major matrix at r * width: present row 0 */
    uint32_t cur[8];
    int ok = level_hash(one, n, H, 0, cur);
    uint32_t idx = index, lvl = 0;
    for (uint32_t size = H / 2; ok && size >= 1; size /= 2, lvl++) {
        uint32_t nxt[8], extra[8];
        if (idx & 1) or_hash_pair(path + lvl * 8, cur, nxt); else or_hash_pair(cur, path + lvl * 8, nxt);
        idx >>= 1;
        if (level_hash(one, n, size, 0, extra)) or_hash_pair(nxt, extra, cur);
        else memcpy(cur, nxt, 32);
    }
    free(one);
    return ok && memcmp(cur, root, 32) == 0 ? 0 : 1;
}

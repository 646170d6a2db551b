// simdjson_amd/csrc/sjgpu_small.hip -- many SMALL documents per launch: one 256-thread workgroup per document.
//
// The reference calls stage1 once per document (dom::parser::parse, ondemand::parser::iterate) or per window
// (document_stream, /root/reference/include/simdjson/dom/document_stream-inl.h:285-317).  For documents of a few KiB the
// tile pipelines of sjgpu_fused.hip are all overhead: a memset, a ticket, descriptors, look-backs between workgroups --
// and three PCIe transfers around them.  Here ONE workgroup walks ONE document 16 KiB at a time (4 waves x one 4 KiB
// chunk), the in-string bit and the output cursor stay in registers, nothing needs clearing, and the buffers may live in
// page-locked HOST memory: the kernel then reads the document and writes the offsets and the 16-byte result across
// PCIe itself, so a call is one launch and one wait (sjgpu_stage1 for documents up to DOCS_SINGLE_MAX bytes), and a
// batch of documents is one launch for all of them (sjgpu_stage1_many / sjgpu_minify_many / sjgpu_validate_utf8_many).
// Same per-chunk scanner, same emission, same UTF-8 list as the large-input kernels: bit-identical output.
#include "sjgpu_device.h"

namespace sjgpu {
namespace {

constexpr u32 DOC_WAVES = 4;
constexpr u32 DOC_STEP_BYTES = DOC_WAVES * CHUNK_BYTES; // 16 KiB per step of the workgroup

// OP 0: stage 1 (out = u32 offsets + the three sentinels), OP 1: minify (out = bytes), OP 2: validate_utf8 (no out)
template <int OP>
__global__ __launch_bounds__(256) void k_docs(const u8 *__restrict__ in_base, const doc_desc *__restrict__ docs, doc_desc single,
                                              void *__restrict__ out_base, scan_result_dev *__restrict__ results) {
  constexpr u32 STAGE_WORDS = (OP == 1) ? (MINIFY_STAGE_BYTES / 4) : EMIT_STAGE_WORDS;
  __shared__ u32 sh_wave[2][DOC_WAVES][4]; // [step parity][wave]: quote parity, count if out, count if in, flags
  __shared__ __attribute__((aligned(16))) u32 sh_stage[(OP == 2) ? 1 : DOC_WAVES][(OP == 2) ? 4 : STAGE_WORDS];
  __shared__ u32 sh_lut[MINIFY_LUT_WORDS];
  __shared__ u32 sh_uq[(OP == 1) ? 1 : DOC_WAVES][(OP == 1) ? 1 : UTF8Q_SLOTS];
  __shared__ u32 sh_flags;

  const doc_desc d = docs ? docs[blockIdx.x] : single;
  const u8 *__restrict__ buf = in_base + d.in_off;
  const u64 len = d.len;
  const u32 wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
  if (threadIdx.x == 0) { sh_flags = 0; }
  if (OP == 1) {
    if (wave == 0) { init_compaction_lut(sh_lut, lane); }
    clear_minify_stage(reinterpret_cast<u8 *>(sh_stage[wave]), lane);
  }
  utf8_queue uq{sh_uq[(OP == 1) ? 0 : wave], 0u, 0u, 0u, buf, len, 0u};
  u32 S = 0, cursor = 0; // workgroup-uniform: in-string bit and output cursor in front of the current step
  u32 *const idx = static_cast<u32 *>(out_base) + d.out_off;   // OP 0
  u8 *const dst = static_cast<u8 *>(out_base) + d.out_off;     // OP 1
  const u64 out_room = (OP == 0) ? u64(d.len) + 3 : u64(d.len);
  bool overflow = false;
  u32 my_flags = 0;
  lds_writes_done();
  __syncthreads();

  for (u64 step_start = 0, step = 0; step_start < len; step_start += DOC_STEP_BYTES, step++) {
    const u32 cur = u32(step) & 1u;
    const u64 cstart = step_start + u64(wave) * CHUNK_BYTES;
    const u64 pos = cstart + u64(lane) * BLOCK_BYTES;
    u64 a = 0, b = 0;
    u32 n_out = 0, n_in = 0, parity = 0, f = 0;
    u32 w[16];
    const bool live = cstart < len; // wave-uniform
    if (live) {
      const u32 lookback = lookback_issue(buf, cstart, lane);
      load_block(buf, pos, len, w);
      wave_carry wc = segment_carry_from(buf, cstart, lane, lookback); // the one kernel that walks: its documents are at most 64 KiB
      if (OP != 1) { uq.pending = utf8_pending_from(lookback, lane); }
      if (OP == 0) {
        const chunk_masks m = scan_chunk<true, true>(w, wc, lane, &uq, u32(cstart / BLOCK_BYTES));
        a = m.cand;
        b = m.string_tail;
        n_out = u32(popc64(a & ~b));
        n_in = u32(popc64(a & b));
        if (__ballot((m.ctrl & m.in_string) != 0)) { f |= 1u; }  // offends if the chunk starts outside a string
        if (__ballot((m.ctrl & ~m.in_string) != 0)) { f |= 2u; } // ... inside
      } else if (OP == 1) {
        const chunk_masks m = scan_chunk<false, false>(w, wc, lane);
        const u64 valid = valid_mask(pos, len);
        a = valid & m.ws;
        b = m.in_string;
        n_out = u32(popc64(valid & ~(a & ~b)));
        n_in = u32(popc64(valid & ~(a & b)));
      } else {
        const planes P = transpose64(w);
        utf8_note_chunk(uq, P, w[15], u32(cstart / BLOCK_BYTES), lane);
      }
      parity = wc.s;
      if (OP != 1) { utf8_drain_if_full(uq, buf, len, false, lane); }
    }
    if (OP == 2) { continue; } // no cross-chunk state at all
    {
      const u32 t_out = wave_sum(n_out), t_in = wave_sum(n_in);
      if (lane == 0) {
        sh_wave[cur][wave][0] = parity;
        sh_wave[cur][wave][1] = t_out;
        sh_wave[cur][wave][2] = t_in;
        sh_wave[cur][wave][3] = f;
      }
    }
    __syncthreads(); // the only barrier of a step: sh_wave is double-buffered by step parity
    u32 s = S, base = cursor, s_end = S, total = cursor;
#pragma unroll
    for (u32 v = 0; v < DOC_WAVES; v++) {
      const u32 q = sh_wave[cur][v][0], o = sh_wave[cur][v][1], i = sh_wave[cur][v][2];
      const u32 add = s_end ? i : o;
      if (v < wave) { base += add; s ^= q; }
      total += add;
      s_end ^= q;
    }
    if (live) {
      if (OP == 0) {
        if (f & (s ? 2u : 1u)) { my_flags |= SJGPU_F_UNESCAPED_CTRL; }
        emit_indices(a & ~(b ^ (s ? ~0ull : 0ull)), u32(pos), lane, idx, out_room, base, sh_stage[wave], overflow);
      } else {
        emit_bytes(w, valid_mask(pos, len) & ~(a & ~(b ^ (s ? ~0ull : 0ull))), lane, dst, base, reinterpret_cast<u8 *>(sh_stage[wave]), sh_lut);
      }
    }
    S = s_end;
    cursor = total;
  }
  if (OP != 1) {
    utf8_drain_rest(uq, buf, len, false, lane);
    if (uq.error) { my_flags |= SJGPU_F_UTF8_ERROR; }
  }
  if (OP == 0 && __ballot(overflow)) { my_flags |= SJGPU_F_IDX_OVERFLOW; }
  if (my_flags && lane == 0) { atomicOr(&sh_flags, my_flags); }
  __syncthreads();
  if (threadIdx.x == 0) {
    scan_result_dev r;
    r.n = (OP == 0) ? cursor : 0u;
    r.flags = sh_flags | (S ? SJGPU_F_UNCLOSED_STRING : 0u);
    r.out_len = (OP == 1) ? ((S && !(d.flags & DOC_KEEP_UNCLOSED)) ? 0ull : u64(cursor)) : 0ull; // json_minifier.h:42-47
    if (OP == 0) { // json_structural_indexer.h:284-286
      idx[cursor] = u32(len);
      idx[cursor + 1] = u32(len);
      idx[cursor + 2] = 0;
    }
    results[blockIdx.x] = r;
  }
}

} // namespace

void launch_docs(int op, const uint8_t *in_base, const doc_desc *docs, doc_desc single, uint32_t count, void *out_base,
                 scan_result_dev *results, hipStream_t stream) {
  if (count == 0) { return; }
  if (op == 0) { hipLaunchKernelGGL(k_docs<0>, dim3(count), dim3(256), 0, stream, in_base, docs, single, out_base, results); }
  else if (op == 1) { hipLaunchKernelGGL(k_docs<1>, dim3(count), dim3(256), 0, stream, in_base, docs, single, out_base, results); }
  else { hipLaunchKernelGGL(k_docs<2>, dim3(count), dim3(256), 0, stream, in_base, docs, single, out_base, results); }
}

} // namespace sjgpu

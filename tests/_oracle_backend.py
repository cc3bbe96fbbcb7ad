"""Test double with the Backend interface, computed by the CPU oracle.  TESTS ONLY: lets the host-side driver,
sharding and merge logic run without a GPU (the product itself has no CPU path)."""
import numpy as np

from oracle import oracle as orc


class OracleBackend:
    def __init__(self, device_id=0):
        self.w = None
        self.dil = None
        self.lm = None
        self.k = 0

    def set_precision(self, mode):
        pass   # the checker computes in float32 either way

    def set_logits(self, mode):
        assert mode in ("f32", 0)

    def set_decode_math(self, mode):
        pass   # the checker IS the host's libm

    def clone_artifacts_from(self, src):
        self.w, self.dil, self.lm, self.k = src.w, src.dil, src.lm, src.k

    def load_weights(self, flat, dilations=(1, 2, 4, 8, 16, 32)):
        self.w = np.asarray(flat, dtype=np.float32)
        self.dil = tuple(dilations)

    def load_lm(self, table, k):
        self.lm, self.k = (None, 0) if table is None else (np.asarray(table, dtype=np.float64), k)

    def forward(self, windows):
        return orc.tcn_forward(self.w, np.asarray(windows, dtype=np.float32), dilations=self.dil)

    def basecall_chunk(self, windows, valid_len, beam_width):
        probs = self.forward(windows)
        return [orc.beam_search_labels(probs[i, : valid_len[i]], beam_width)[0] for i in range(probs.shape[0])]

    def basecall_global(self, windows, read_win_off, pads, step, beam_width, use_lm, s_threshold=0.0, r_threshold=0.0):
        probs = self.forward(windows)
        out = []
        for r in range(len(pads)):
            m = orc.assemble_matrices(probs[read_win_off[r]:read_win_off[r + 1]], int(pads[r]), step)
            try:
                lab, _ = orc.beam_search_labels(m, beam_width, self.lm if use_lm else None, s_threshold, r_threshold,
                                                self.k if use_lm else 0)
            except KeyError:
                lab = None    # (what radian_amd.Backend returns for RD_LEN_MISSING_CONTEXT: a sparse LM's absent context was reached)
            out.append(lab)
        return out

    def _windows(self, sig, chunk, step):
        w, pad = orc.get_windows(np.asarray(sig, dtype=np.float32), chunk, step)
        valid = np.full(w.shape[0], chunk, dtype=np.int32)
        valid[-1] = chunk - pad
        return w.astype(np.float32), valid, pad

    def basecall_reads_chunk(self, signals, chunk_len, step, beam_width):
        out = []
        for sig in signals:
            w, valid, _ = self._windows(sig, chunk_len, step)
            out.append(self.basecall_chunk(w, valid, beam_width))
        return out

    def basecall_reads_global(self, signals, chunk_len, step, beam_width, use_lm, s_threshold=0.0, r_threshold=0.0):
        out = []
        for sig in signals:
            w, _, pad = self._windows(sig, chunk_len, step)
            out.append(self.basecall_global(w, np.array([0, w.shape[0]]), np.array([pad]), step, beam_width, use_lm,
                                            s_threshold, r_threshold)[0])
        return out

    def _normalise(self, raws, clip):
        sigs, status = [], []
        for raw in raws:
            try:
                sigs.append(orc.mad_normalise(np.asarray(raw), clip).astype(np.float32))
                status.append(0)
            except ValueError as e:
                sigs.append(np.zeros(len(raw), dtype=np.float32))
                status.append(1 if "MAD" in e.args[0] else 2)
        return sigs, np.array(status, dtype=np.int32)

    def basecall_raw_chunk(self, raws, outlier_clip, chunk_len, step, beam_width):
        sigs, status = self._normalise(raws, outlier_clip)
        return self.basecall_reads_chunk(sigs, chunk_len, step, beam_width), status

    def basecall_raw_global(self, raws, outlier_clip, chunk_len, step, beam_width, use_lm, s_threshold=0.0, r_threshold=0.0):
        sigs, status = self._normalise(raws, outlier_clip)
        return self.basecall_reads_global(sigs, chunk_len, step, beam_width, use_lm, s_threshold, r_threshold), status

    def close(self):
        pass

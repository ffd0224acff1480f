"""CPU oracle for the DDSP synthesis hot path.

TEST INFRASTRUCTURE ONLY.  This package is a CPU restatement (PyTorch CPU ops +
numpy) of the reference algorithm (tarepan/DDSP-SVC-official, `ddsp/core.py`,
`ddsp/vocoder.py:372-550`, `ddsp/unit2control.py`, `ddsp/pcmer.py`,
`ddsp/loss.py`, `gui.py:405-430`, `main.py:50-57,111-116`).  Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import
it, and only as the checker / reported baseline.  The product path
(`ddsp-svc-official_amd/`) never imports it and has no CPU fallback.

Pinning status (see DESIGN.md "Oracle"):
  * `oracle.dsp`   - pinned by golden vectors generated from the unmodified
                     reference module `ddsp/core.py` (tests/golden/make_golden.py)
                     and by the reference's own known-answer tests
                     (`ddsp/core.py:54-97`).
  * `oracle.synth` - DSP composition pinned the same way (golden G6/G7 from
                     `ddsp/vocoder.py` imported with placeholders for absent
                     third-party packages).
  * `oracle.ctrlnet` - parity UNPINNED at the `extorch` boundary
                     (`Conv1dEx`/`Transpose` are a third-party package that is
                     not installed; the c=False reading `nn.Conv1d(padding="same")`
                     / `x.transpose(1, 2)` is this build's contract).
  * `oracle.loss`  - parity UNPINNED at the `torchaudio.Spectrogram` boundary
                     (torchaudio absent; restated from its documented semantics).
"""

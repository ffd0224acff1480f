"""API surface of the reference's `enhancer.py` (`Enhancer(type, ckpt, device).enhance(...)`).

The NSF-HiFiGAN post-net is OUT OF SCOPE of this build (it is the step after the hot path, needs
a pretrained checkpoint that is not available offline, and is a different model family).  The class keeps the
constructor and `enhance` signature so callers written against the reference import and fail loudly - never
silently skip - when they ask for it.
"""


class Enhancer:
    def __init__(self, enhancer_type, enhancer_ckpt, device=None):
        if enhancer_type != "nsf-hifigan":
            raise ValueError(f" [x] Unknown enhancer: {enhancer_type}")     # reference enhancer.py:18
        raise NotImplementedError(
            "the NSF-HiFiGAN enhancer is not part of the MI355X synthesis path (SURVEY 8f, next-in-line component); "
            "run the callers with the enhancer disabled (main.py -e false)")

    def enhance(self, audio, sample_rate, f0, hop_size, adaptive_key=0, silence_front=0):
        raise NotImplementedError

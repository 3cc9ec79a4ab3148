"""calitas_amd -- MI355X-native CALITAS SearchReference hot path (HIP kernels behind a C ABI)."""
from .aligner import (Alignment, CalitasError, Context, Defaults, Guide, SearchReference, make_params, read_hits,  # noqa: F401
                      window_filter)
from .tools import GuideAlignment, SequentialGuideAligner, align_to_reference, pairwise_align_sequences, read_fasta  # noqa: F401,E402
from .variants import prepare_vcf, read_vcf  # noqa: F401,E402

"""Inputs of the read path that the reference keeps in ocr_agent/config.py:16-36 — model id, pixel bounds, token budget,
prompt, loop thresholds and the strategy table — under the same names, for hosts without the reference package.
The values are the reference's (they decide which reads run and what the processor sees); tests/golden/nodes_kats.json and
preprocess_kats.json were generated with them."""
OLMOCR_MODEL = "allenai/olmOCR-2-7B-1025"
OCR_PROMPT = "Extract and return all the text from this handwritten document."
OCR_MIN_PIXELS, OCR_MAX_PIXELS = 256 * 256, 1024 * 1024
OCR_MAX_NEW_TOKENS = 2048

MAX_ITERATIONS, PLATEAU_PATIENCE = 10, 2
ACCEPT_THRESHOLD, AGREEMENT_THRESHOLD = 85, 80

# Chains of transforms, tried top to bottom: the first two are the initial reads, the third the tie-breaker, the rest re-OCR
# rounds.  The last chain repeats the first one's label, so it can never run (nodes.py:36-39 dedups reads by label).
PREPROCESSING_STRATEGIES = [chain.split("+") for chain in (
    "deskew+high_contrast+binarize",
    "high_contrast+binarize",
    "deskew+high_contrast+sharpen",
    "deskew+denoise+high_contrast",
    "deskew+remove_lines+high_contrast",
    "deskew+high_contrast+binarize",
)]

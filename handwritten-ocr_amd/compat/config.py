"""Tunable constants of the OCR path, same names and values as ocr_agent/config.py:16-36 (they are inputs of the
path: model id, pixel bounds, token budget, prompt, thresholds, strategy table)."""
OLMOCR_MODEL = "allenai/olmOCR-2-7B-1025"
OCR_MAX_PIXELS = 1024 * 1024
OCR_MIN_PIXELS = 256 * 256
OCR_MAX_NEW_TOKENS = 2048
OCR_PROMPT = "Extract and return all the text from this handwritten document."

MAX_ITERATIONS = 10
ACCEPT_THRESHOLD = 85
PLATEAU_PATIENCE = 2
AGREEMENT_THRESHOLD = 80
# tried top to bottom: the first two are the initial reads, the third the tie-breaker, the rest re-OCR rounds.
# Entries 0 and 5 share a label, so the last one can never run (nodes.py:36-39 dedups by label).
PREPROCESSING_STRATEGIES = [
    ["deskew", "high_contrast", "binarize"],
    ["high_contrast", "binarize"],
    ["deskew", "high_contrast", "sharpen"],
    ["deskew", "denoise", "high_contrast"],
    ["deskew", "remove_lines", "high_contrast"],
    ["deskew", "high_contrast", "binarize"],
]

"""MI355X-native implementation of the multi-strategy page-read path of marwanbounassif/handwritten-ocr.

Import as ``handwritten_ocr_amd`` (alias package at the repo root).  Sub-modules:
  build        compile the HIP / C++ libraries in-tree (gfx950)
  _lib         ctypes binding of include/hwocr.h (fails loudly when the HIP library is missing)
  text         native compare / merge / CER (mirror of ocr_agent.tools string functions)
  imageproc    smart_resize + page -> device pixels (mirror of the HF Qwen2-VL image processor)
  preprocess   the preprocessing strategies (mirror of ocr_agent.tools.preprocess_image)
  engine       Qwen2-VL-family read engine (vision tower, prefill, graph-captured decode)
  tools        drop-in surface: run_ocr, preprocess_image, unload_ocr_model, compare_versions, merge_versions
  compat       ocr_agent-shaped state / nodes / routing for hosts without langgraph / ollama
  shard        one-process-per-GPU page sharding + RCCL gather of token streams
"""
__version__ = "0.1.0"

"""ORACLE — test infrastructure only.

CPU restatements of the reference's algorithms for the page-read path, each citing the reference file:line it
follows.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import from here; the product
package (handwritten-ocr_amd/) never does.
"""

"""`ocr_agent`-shaped host layer for machines that have neither the reference package nor langgraph / ollama
(the GPU box): configuration constants, the OCRState contract, the two hot-path graph nodes and a small
state-machine runner with scripted critic / editor / arbitrator stand-ins.  On a user site that has the reference
installed none of this is needed — `handwritten_ocr_amd.tools.install()` patches the engine into `ocr_agent`."""

"""`import faiss.contrib.torch_utils` (finetune_jpq.py:9) makes Faiss indexes accept torch tensors; `PQIndex.search` does so
natively (CUDA tensors in -> CUDA tensors out), so there is nothing to patch."""

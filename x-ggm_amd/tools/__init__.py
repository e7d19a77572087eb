"""host-side input pipeline of the hot path (reference: src/tools/, src/vqa/vqacpv2_data.py)"""

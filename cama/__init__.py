"""`import cama...` resolves to the MI355X-native implementation in cama_amd/ so that the reference's
main.py runs unchanged against this repository."""

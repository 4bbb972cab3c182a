"""CPU oracle for the BiddingSimulation step path - TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; the product
package (adcraft_amd) must never import it.
"""

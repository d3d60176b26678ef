"""Mirror of the reference's modules/ package for the Markov set-abstraction path."""

"""
Entry script, same surface as the reference's main.py (:44-451): module constants,
setup_problem(), perform_inference(problem, data_format, nsamples), main().
Run as `python main.py` from this directory (like the reference) or import it from the package.
"""
if __package__:
    from .RateStateModel import RateStateModel
    from .RSF import RSF
else:  # `python main.py` from this directory, or this directory on sys.path: the reference's flat imports (main.py:44-46)
    from RSF import RSF
    from RateStateModel import RateStateModel

# main.py:50-56
NUMBER_SLIP_VALUES = 5
LOWEST_SLIP_VALUE = 100.0
LARGEST_SLIP_VALUE = 5000.0
QSTART = 1000.0
QPRIORS = ["Uniform", 0.0, 10000.0]
NUMBER_TIME_STEPS = 500
NSAMPLES = 500


def setup_problem():
    problem = RSF(number_slip_values=NUMBER_SLIP_VALUES, lowest_slip_value=LOWEST_SLIP_VALUE,
                  largest_slip_value=LARGEST_SLIP_VALUE, qstart=QSTART, qpriors=QPRIORS)
    problem.model = RateStateModel(number_time_steps=NUMBER_TIME_STEPS)
    problem.data = problem.generate_time_series()
    return problem


def perform_inference(problem, data_format, nsamples):
    problem.format = data_format
    return problem.inference(nsamples=nsamples)


def main():
    problem = setup_problem()
    json_time = perform_inference(problem, "json", NSAMPLES)
    print(f"inference wall time: {json_time:.2f} s")
    try:
        import matplotlib.pyplot as plt

        plt.show()
        plt.close("all")
    except Exception:
        pass


if __name__ == "__main__":
    main()
